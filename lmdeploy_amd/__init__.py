"""MI355X-native TurboMind quantized-inference hot path behind the lmdeploy.pipeline() surface."""
__version__ = '0.1.0'

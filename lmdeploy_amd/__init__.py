"""MI355X-native TurboMind quantized-inference hot path behind the lmdeploy.pipeline() surface.

    from lmdeploy_amd import pipeline, TurbomindEngineConfig, GenerationConfig
    pipe = pipeline('synthetic:llama3_8b', backend_config=TurbomindEngineConfig(quant_policy=8, session_len=2048))
"""
from .api import pipeline
from .messages import GenerationConfig, QuantPolicy, Response, ResponseType, TurbomindEngineConfig
from .pipeline import Pipeline

__version__ = '0.1.0'
__all__ = ['pipeline', 'Pipeline', 'TurbomindEngineConfig', 'GenerationConfig', 'QuantPolicy', 'Response',
           'ResponseType']

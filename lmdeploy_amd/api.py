"""`lmdeploy.pipeline()` (reference: lmdeploy/api.py:15-82) -- same signature, MI355X engine behind it."""
from __future__ import annotations

from .messages import TurbomindEngineConfig
from .pipeline import Pipeline


def pipeline(model_path: str,
             backend_config: TurbomindEngineConfig | None = None,
             chat_template_config=None,
             log_level: str = 'WARNING',
             max_log_len: int | None = None,
             trust_remote_code: bool = False,
             speculative_config=None,
             allowed_media_domains=None,
             **kwargs) -> Pipeline:
    """Create a pipeline for inference.  `model_path` is an HF checkpoint directory (Llama / InternLM2, AWQ W4A16
    group 128 or fp16) or `synthetic:<llama3_8b|internlm2_1_8b|internlm2_20b|llama3_70b|tiny>` for random weights
    with the real shapes."""
    if allowed_media_domains is not None:
        raise NotImplementedError('multimodal inputs are outside the MI355X hot path')
    return Pipeline(model_path, backend_config=backend_config, chat_template_config=chat_template_config,
                    speculative_config=speculative_config, **kwargs)

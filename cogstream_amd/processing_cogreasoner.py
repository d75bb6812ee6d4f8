"""The reference's plugin name for the processor: `preprocessor_config.json` -> auto_map ->
`processing_cogreasoner.Videollama3Qwen2Processor` (model/preprocessor_config.json:2-5), what
`AutoProcessor.from_pretrained(model_path, trust_remote_code=True)` resolves to in evaluate/answer_generate.py:179."""
from __future__ import annotations

from .processing import CogStreamProcessor


class Videollama3Qwen2Processor(CogStreamProcessor):
    """model/processing_cogreasoner.py:732-744 (`__call__(conversation=..., add_system_prompt, add_generation_prompt,
    return_tensors)`), `.tokenizer`, `.batch_decode` -- for in-memory clips (the file-decoding front end needs ffmpeg,
    cogstream_amd/video_io.py)."""


__all__ = ["Videollama3Qwen2Processor"]

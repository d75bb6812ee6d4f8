"""The three factory names evaluate/answer_generate.py imports from transformers / peft (:16-17,173-183), resolved
the way `trust_remote_code` does it: the checkpoint directory's own `auto_map` names the module and class
(model/config.json:6-9, model/preprocessor_config.json:2-5) and they are looked up IN THIS PACKAGE -- so the driver's

    model = AutoModelForCausalLM.from_pretrained(path, trust_remote_code=True, torch_dtype=torch.bfloat16,
                                                 attn_implementation="flash_attention_2")
    processor = AutoProcessor.from_pretrained(path, trust_remote_code=True)
    model = PeftModel.from_pretrained(model, adapter_1, adapter_name="full_module")
    model.load_adapter(adapter_2, adapter_name="language_module")
    model.to(local_rank)

runs unchanged after `from cogstream_amd.auto import AutoModelForCausalLM, AutoProcessor, PeftModel`."""
from __future__ import annotations

import importlib
import json
import os

_DEFAULT_MODEL = "cogreasoner_chat.Videollama3Qwen2ForCausalLM"
_DEFAULT_PROCESSOR = "processing_cogreasoner.Videollama3Qwen2Processor"


def _resolve(path: str, json_name: str, key: str, default: str):
    target = default
    p = os.path.join(path, json_name)
    if os.path.exists(p):
        with open(p, encoding="utf-8") as f:
            target = (json.load(f).get("auto_map") or {}).get(key, default)
    module, _, cls = target.rpartition(".")
    try:
        mod = importlib.import_module(f"{__package__}.{module}")
    except ModuleNotFoundError as e:
        raise ValueError(f"{p}: auto_map[{key!r}] = {target!r} names a module this package does not provide") from e
    if not hasattr(mod, cls):
        raise ValueError(f"{p}: auto_map[{key!r}] = {target!r}: no class {cls} in {mod.__name__}")
    return getattr(mod, cls)


class AutoModelForCausalLM:
    @staticmethod
    def from_pretrained(pretrained_model_name_or_path: str, **kwargs):
        cls = _resolve(pretrained_model_name_or_path, "config.json", "AutoModelForCausalLM", _DEFAULT_MODEL)
        return cls.from_pretrained(pretrained_model_name_or_path, **kwargs)


class AutoProcessor:
    @staticmethod
    def from_pretrained(pretrained_model_name_or_path: str, **kwargs):
        cls = _resolve(pretrained_model_name_or_path, "preprocessor_config.json", "AutoProcessor", _DEFAULT_PROCESSOR)
        return cls.from_pretrained(pretrained_model_name_or_path, **kwargs)


class PeftModel:
    @staticmethod
    def from_pretrained(model, model_id: str, adapter_name: str = "default", **unused):
        """peft.PeftModel.from_pretrained(model, path, adapter_name=...) (:181): loads the adapter into `model` and makes
        it the active one (peft's behaviour); returns the same object, which keeps `load_adapter` / `set_adapter`"""
        model.load_adapter_from_path(model_id, adapter_name)
        model.set_adapter(adapter_name)
        return model


__all__ = ["AutoModelForCausalLM", "AutoProcessor", "PeftModel"]

"""The reference's plugin name for the model: `config.json` -> auto_map -> `cogreasoner_chat.Videollama3Qwen2ForCausalLM`
(model/config.json:6-9), the class `AutoModelForCausalLM.from_pretrained(..., trust_remote_code=True)` resolves to in
evaluate/answer_generate.py:173-178. Here it is the HIP-backed model of cogstream_amd.chat under that name, with the
peft call forms the reference driver uses on it (:181-182)."""
from __future__ import annotations

from .chat import CogReasoner


class Videollama3Qwen2ForCausalLM(CogReasoner):
    """model/cogreasoner_chat.py:587-908 (`qa_selection`, `generate`, `set_adapter`, `.device`, `.to`, `.eval`)"""

    def load_adapter(self, *args, adapter_name: str = "default", **kw) -> None:
        """peft: `model.load_adapter(path, adapter_name="language_module")` (evaluate/answer_generate.py:182). A single
        directory argument is a peft adapter directory; the state-dict form of CogReasoner.load_adapter is kept."""
        if len(args) == 1 and isinstance(args[0], (str, bytes)) or (not args and "model_id" in kw):
            path = args[0] if args else kw.pop("model_id")
            return self.load_adapter_from_path(path, adapter_name)
        if "adapter_name" not in kw and len(args) < 4:
            kw["adapter_name"] = adapter_name
        return super().load_adapter(*args, **kw)


__all__ = ["Videollama3Qwen2ForCausalLM"]

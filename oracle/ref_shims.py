"""Import the *reference's own* Python modules from /root/reference in THIS container.

TEST INFRASTRUCTURE (used only by oracle/make_golden.py to pin the oracle).  /root/reference
does not exist on the GPU box; nothing under tests/, bench.py or smoke() imports this file.
No reference source is copied: the modules are imported in place, with in-memory stubs for
the third-party packages that are not installed here (timm, flash_attn, torchvision, decord,
av) -- SURVEY.md Appendix B.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns a namespace with the reference modules (llava_next_video, clip, iv2, phi3, llama, template, video_utils)."""
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import torch
    import torch.nn as nn
    import transformers  # noqa: F401  (must precede the stubs)
    from transformers import CLIPVisionConfig, LlamaConfig  # noqa: F401
    import transformers.models.llava.modeling_llava  # noqa: F401

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    if "timm" not in sys.modules:
        _stub("timm")
        _stub("timm.models")
        _stub("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=torch.nn.init.trunc_normal_)
    if "flash_attn" not in sys.modules:
        _stub("flash_attn")
        _stub("flash_attn.flash_attn_interface", flash_attn_varlen_qkvpacked_func=None)
        _stub("flash_attn.bert_padding", unpad_input=None, pad_input=None)
    if "torchvision" not in sys.modules:
        _stub("torchvision")
        dummy = type("Dummy", (), {})
        _stub("torchvision.transforms", Normalize=dummy, Compose=dummy, InterpolationMode=dummy, ToTensor=dummy,
              Resize=dummy, CenterCrop=dummy, ToPILImage=dummy)
    if "av" not in sys.modules:
        _stub("av")
    if "decord" not in sys.modules:
        bridge = types.SimpleNamespace(set_bridge=lambda *_a, **_k: None)
        _stub("decord", VideoReader=object, DECORDError=Exception, bridge=bridge)
    # the installed HF `datasets` shadows the reference's namespace package: register stubs and load by path
    _stub("datasets")
    _stub("datasets.chat")
    spec = importlib.util.spec_from_file_location("datasets.chat.base_template", REF + "/datasets/chat/base_template.py")
    tmpl = importlib.util.module_from_spec(spec)
    sys.modules["datasets.chat.base_template"] = tmpl
    spec.loader.exec_module(tmpl)

    ns = types.SimpleNamespace()
    ns.template = tmpl
    ns.llava = importlib.import_module("models.llava_next_video")
    ns.clip = importlib.import_module("models.modeling_clip")
    ns.iv2 = importlib.import_module("models.internvideo2")
    ns.phi3 = importlib.import_module("models.modeling_phi3")
    ns.llama = importlib.import_module("models.modeling_llama")
    ns.video_utils = importlib.import_module("mm_utils.video_utils")
    old_argv = sys.argv
    sys.argv = ["inference.py"]
    try:
        ns.inference = importlib.import_module("inference")
    finally:
        sys.argv = old_argv
    return ns

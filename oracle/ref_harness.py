"""TEST INFRASTRUCTURE ONLY -- container-side loader for the upstream reference.

Imports /root/reference (read-only, never copied) on CPU so that golden vectors can be
minted (oracle/make_golden.py) and the CPU restatement (oracle/teramind_cpu.py) can be
validated against the real thing.  The reference does not exist on the GPU box, so nothing
under tests/ -m gpu, bench.py or __graft_entry__.smoke() may import this module.

Stub recipe follows SURVEY.md section 8c: three import-time-only dependencies are absent in
this image (tkinter/turtle, torchvision, timm) plus zarr/sparse pulled in by config.py.
The only *arithmetic* living in a stubbed module is timm==1.0.14's `Mlp`
(reference call site model/MBAblocks.py:461); it is restated here as
fc1 -> act -> drop1 -> norm(Identity) -> fc2 -> drop2.  No reference test pins it:
"parity unpinned" at that one boundary (key names mlp.fc1/fc2 and tanh-GELU are fixed by
model/MBAblocks.py:18,461).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("TERAMIND_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


def _install_stubs():
    import torch.nn as nn

    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    if "turtle" not in sys.modules:
        mod("turtle", forward=lambda *a, **k: None)
    try:
        import torchvision  # noqa: F401
    except Exception:
        mod("torchvision")
        mod("torchvision.models")
        mod("torchvision.models.feature_extraction",
            create_feature_extractor=lambda *a, **k: None)
        mod("torchvision.transforms")
        mod("torchvision.transforms.functional")
        sys.modules["torchvision"].models = sys.modules["torchvision.models"]
        sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
        sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
        sys.modules["torchvision.models"].feature_extraction = sys.modules["torchvision.models.feature_extraction"]
    for name in ("zarr", "sparse"):
        try:
            __import__(name)
        except Exception:
            mod(name)

    try:
        from timm.models.vision_transformer import Mlp  # noqa: F401
    except Exception:
        class Mlp(nn.Module):
            """Restatement of timm 1.0.14 layers/mlp.py::Mlp (the fields the reference uses)."""

            def __init__(self, in_features, hidden_features=None, out_features=None,
                         act_layer=nn.GELU, norm_layer=None, bias=True, drop=0.0,
                         use_conv=False):
                super().__init__()
                out_features = out_features or in_features
                hidden_features = hidden_features or in_features
                self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
                self.act = act_layer()
                self.drop1 = nn.Dropout(drop)
                self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
                self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
                self.drop2 = nn.Dropout(drop)

            def forward(self, x):
                x = self.fc1(x)
                x = self.act(x)
                x = self.drop1(x)
                x = self.norm(x)
                x = self.fc2(x)
                x = self.drop2(x)
                return x

        mod("timm")
        mod("timm.models")
        mod("timm.models.vision_transformer", Mlp=Mlp)


_loaded = None


def load():
    """Returns a namespace with the reference's `model`, `diffusion`, `prep_config_parm`,
    `GenerativeType` and `unet_attn` modules, imported from REF_ROOT on CPU."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    sys.dont_write_bytecode = True
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        import model as ref_model            # noqa
        import diffusion as ref_diffusion    # noqa
        from config_parm import prep_config_parm
        from utils.choices import GenerativeType
        from model import unet_attn as ref_unet_attn
    ns = types.SimpleNamespace(model=ref_model, diffusion=ref_diffusion,
                               prep_config_parm=prep_config_parm,
                               GenerativeType=GenerativeType, unet_attn=ref_unet_attn)
    _loaded = ns
    return ns


def make_conf(bat=1, size=64, stain="all", mouse="638850", nrna=229, srna=4, method="ours",
              net_ch=None):
    import contextlib
    import io
    ns = load()
    with contextlib.redirect_stdout(io.StringIO()):
        conf = ns.prep_config_parm("", bat, size, 1, stain, mouse, nrna, srna,
                                   method=method, is_test=True)
    if net_ch is not None:
        conf.net_ch = net_ch
    return conf


def make_model(conf):
    return conf.make_model_conf().make_model().eval()


def make_sampler(conf, T, gen_type="ddim"):
    ns = load()
    conf.beatgans_gen_type = getattr(ns.GenerativeType, gen_type)
    return conf._make_diffusion_conf(T).make_sampler()

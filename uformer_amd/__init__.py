"""uformer_amd -- MI355X-native (gfx950) Uformer LeWin-block hot path behind the reference's
``Uformer(nn.Module)`` boundary.  See DESIGN.md / INTEGRATION.md and include/uformer_hip.h."""
from .spec import UformerConfig, arch_config, state_dict_spec, synth_input, synth_state_dict  # noqa: F401

__all__ = ["Uformer", "get_arch", "UformerConfig", "arch_config", "state_dict_spec", "synth_state_dict", "synth_input"]


def __getattr__(name):  # lazy: importing the package must not require torch.nn module build
    if name in ("Uformer", "get_arch", "LeWinTransformerBlock", "WindowAttention", "LeFF", "Downsample", "Upsample",
                "InputProj", "OutputProj", "BasicUformerLayer", "window_partition", "window_reverse"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)

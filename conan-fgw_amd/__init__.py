"""conan-fgw_amd: MI355X-native hot path of ConAN-FGW (SchNet/ViSNet message passing + FGW barycenter).

Host side mirrors the reference's model interface (conan_fgw/src/model/graph_embeddings/schnet_no_sum.py,
visnet.py and conan_fgw/src/model/fgw/barycenter.py); compute is hand-written HIP for gfx950 behind the
C-ABI declared in include/conan_fgw_hip.h.  There is no CPU fallback: ops raise if the HIP library is
missing or a tensor is not on a GPU.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # `conan_fgw_amd.get_model(name, device, **kw)` / `conan_fgw_amd.EquivModelsHolder`: the reference's model factory
    # (conan_fgw/src/model/common.py:469-546), resolved lazily so that importing the package needs neither torch nor the .so
    if name in ("get_model", "EquivModelsHolder"):
        from . import head
        return getattr(head, name)
    raise AttributeError(name)

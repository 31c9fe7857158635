"""MI355X-native per-frame reconstruction hot path of pk17r/online_3d_reconstruction.

The product is libo3dr.so (hand-written HIP for gfx950 behind the C ABI of include/o3dr.h).  This
package is the thin Python host layer over that ABI: `Context` mirrors the four `Pose` member
functions of the reference (pose.h:198,199,216,231) plus the fan-out/accumulate loop, `synth`
generates the benchmark inputs of SURVEY.md section 8d, `dist` shards frames over ranks.

There is no CPU fallback: importing works anywhere, but creating a `Context` without the built
library or without a GPU raises.
"""
from ._lib import POINT, O3drError, lib_path, load_library  # noqa: F401
from .api import Context, Params  # noqa: F401

__all__ = ["Context", "Params", "POINT", "O3drError", "lib_path", "load_library"]

"""MI355X-native MPPI rollout engine behind the Python call surface of SokhengDin/DNN-MPPI-MPC's
MPPI controllers.  The hot path (sample -> rollout -> cost -> softmin weight -> reduce) is
hand-written HIP for gfx950 in lib/libmppi_hip.so (C ABI: include/mppi_hip.h).

    from dnn_mppi_mpc_amd import MPPIAlgorithms, MPPIRacecarController

(the directory name has a hyphen; the top-level ``dnn_mppi_mpc_amd`` module aliases it).
"""
from . import _capi
from ._capi import MppiError, load_library
from .build import build as build_library, source_id
from .controllers import MPPIAlgorithms, MPPIRacecarController
from .engine import Engine
from . import callback_mppi, paths

__all__ = ["MPPIAlgorithms", "MPPIRacecarController", "Engine", "MppiError", "load_library", "build_library", "paths", "callback_mppi"]

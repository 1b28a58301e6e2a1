"""CPU: the C-ABI library loads and exports every symbol include/mppi_hip.h declares, the ctypes
prototypes cover exactly that set, and without a GPU the product fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mppi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mppi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("mppi_create", "mppi_destroy", "mppi_step", "mppi_set_ref_path", "mppi_set_obstacles",
                 "mppi_step_begin", "mppi_step_end", "mppi_get_costs", "mppi_rollout_viz", "mppi_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi
    pkg.build_library()
    lib = C.CDLL(_capi.LIB_PATH)
    names = declared_functions()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mppi_hip.h but not exported"
    assert sorted(_capi.PROTOTYPES) == names  # the ctypes binding covers the header exactly
    header = open(os.path.join(ROOT, "include", "mppi_hip.h")).read()
    declared = int(re.search(r"#define\s+MPPI_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.mppi_abi_version() == declared == _capi.ABI_VERSION


def test_driver_entry_build_checks_the_current_abi():
    """__graft_entry__.build() must compare against the mirror's ABI_VERSION, not a literal (it asserted version 1 for
    a while after the ABI had moved to 2 and failed the driver's build check)."""
    import inspect

    import __graft_entry__ as entry
    src = inspect.getsource(entry.build)
    assert "ABI_VERSION" in src and not re.search(r"mppi_abi_version\(\)\s*==\s*\d", src)


def test_config_struct_matches_header_size():
    """sizeof(mppi_config) as the C compiler sees it == the ctypes mirror (struct_size handshake)."""
    import subprocess
    import tempfile
    from dnn_mppi_mpc_amd import _capi
    src = ('#include <stdio.h>\n#include "mppi_hip.h"\nint main(){printf("%zu %zu %zu", sizeof(mppi_config), sizeof(mppi_stats), '
           'sizeof(mppi_cb_config));}')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        a, b, cb = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert int(a) == C.sizeof(_capi.MppiConfig)
    assert int(b) == C.sizeof(_capi.MppiStats)
    from dnn_mppi_mpc_amd import callback_mppi
    assert int(cb) == C.sizeof(callback_mppi.MppiCbConfig)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less hosts")
    import numpy as np

    import dnn_mppi_mpc_amd as pkg
    with pytest.raises(pkg.MppiError) as e:
        pkg.MPPIAlgorithms(delta_t=0.1, ref_path=np.zeros((10, 3)), max_speed=1.0, max_omega=1.0, num_samples_K=8,
                           num_horizons_T=10, param_exploration=0.1, param_lambda=1.0, param_alpha=0.5,
                           sigma=np.eye(2), stage_cost_weight=np.ones(3), terminal_cost_weight=np.ones(3))
    assert e.value.code == -3  # MPPI_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import it."""
    pkg_dir = os.path.join(ROOT, "dnn-mppi-mpc_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "mppi_oracle" not in text, f

"""Import alias: ``import dnn_mppi_mpc_amd`` loads the package directory ``dnn-mppi-mpc_amd/``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("dnn-mppi-mpc_amd")

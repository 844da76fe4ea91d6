"""Import-time stub of the (absent) pandapower package.

TEST INFRASTRUCTURE ONLY. pandapower is not installed in the build image and there is no
network; the reference package ``grid2op`` (mounted read-only at /root/reference) imports it at
module import time (grid2op/Backend/pandaPowerBackend.py:18). This stub only lets those imports
succeed so that the *rest* of the unmodified reference (Environment, Runner, Observation,
the backend conformance kit) can drive ``grid2op_amd.HipBackend``. It implements no arithmetic:
any attempt to actually run pandapower raises.
"""
__version__ = "3.1.1"


class _Powerflow:
    class LoadflowNotConverged(Exception):
        pass


powerflow = _Powerflow()


def __getattr__(name):
    raise AttributeError(
        f"pandapower stub: '{name}' is not available (pandapower is not installed in this image)"
    )

"""Development switches, behind ONE gate.

The shipped dispatch (ops.py, kpconv.py, regtr.py, _lib.py) has A/B switches that were used to measure alternatives (DESIGN.md,
docs/NEGATIVES.md).  They are read from the environment ONLY when REGTR_DEV=1 is set as well; without it every switch is its
default, so a stray REGTR_* variable in a production environment cannot re-route a launch (tests/test_cabi.py:
test_stray_env_switch_does_not_reroute).  The C side has the same gate at compile time (csrc/gemm_x3.hip: REGTR_DEV_ENV)."""
import os

DEV = os.environ.get('REGTR_DEV', '') == '1'


def flag(name, default):
    """Value of the environment switch `name` in a development process (REGTR_DEV=1), `default` otherwise."""
    return os.environ.get(name, default) if DEV else default


def on(name, default='1'):
    return flag(name, default) != '0'

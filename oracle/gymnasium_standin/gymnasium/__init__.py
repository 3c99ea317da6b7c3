"""Minimal stand-in for the `gymnasium` package -- TEST / ORACLE INFRASTRUCTURE ONLY.

gymnasium is not installed in the build image and there is no network.  The
reference (gym-electric-motor, pure Python) only needs the handful of names
below to import and run (`Box/Discrete/MultiDiscrete/Tuple`, `Env`, `register`,
`make`).  This stand-in lets `oracle/make_golden.py` execute the UNMODIFIED
reference from /root/reference to record golden vectors.  Nothing in the
product package imports it.
"""
from . import spaces  # noqa: F401
from .core import Env, Wrapper  # noqa: F401
from .envs.registration import make, register, registry  # noqa: F401

__version__ = "1.0.0-standin"

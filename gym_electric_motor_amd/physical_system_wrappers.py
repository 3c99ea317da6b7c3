"""Parameter-holder mirrors of the two ACTION-side physical-system wrappers that sit directly in front of `simulate()`
(SURVEY.md section 8f rank 2).  They contain no logic: `make(..., physical_system_wrappers=(...))` (or the
`action_delay=` / `action_frame=` arguments of `BatchedSCMLSystem`) folds them into the kernel's action stage, so the
fused rollout stays on the device.

    DeadTimeProcessor(steps[, reset_action]) physical_system_wrappers/dead_time_processor.py:8-85
    DqToAbcActionProcessor.make(motor_type)  physical_system_wrappers/dq_to_abc_action_processor.py:9-175

As in the reference, wrappers are applied innermost first: `(DeadTimeProcessor(2), DqToAbcActionProcessor.make("PMSM"))`
delays the abc action by two steps and lets the dq processor advance its angle by 0.5 + 2 steps (lines 83-86).
Observation-side wrappers (flux observer, cos/sin, current sum, noise) are post-processing and stay on the host.
"""


class DeadTimeProcessor:
    """The converter receives the action submitted `steps` control steps earlier; every reset refills the queue with the
    reset actions: zeros (the reference's default), or what a custom `reset_action` callable returns (dead_time_processor.py:27-50:
    a list of `steps` actions of the wrapped system's action space) -- on the accelerated path those `steps` actions must be ONE
    and the same action, which travels as a per-handle constant (gemx_config.action_delay_reset); a list of different actions is
    refused with a message."""

    def __init__(self, steps=1, reset_action=None, physical_system=None):
        self._steps = int(steps)
        assert self._steps > 0, f'The number of steps has to be greater than 0. A "{steps}" has been passed.'
        self._reset_actions = reset_action

    @property
    def dead_time(self):
        return self._steps


def _reset_action_row(w):
    """The ONE action a DeadTimeProcessor's custom reset_action refills the queue with, as a flat list of numbers (a MultiDiscrete
    action stays [a0, a1]: the system flattens it), or None for the default zeros.  Works on this module's holder and on the
    reference's own instance (whose set_physical_system() wraps the default into a callable as well: zeros come back as zeros)."""
    import numpy as np

    fn = getattr(w, "_reset_actions", None)
    if fn is None:
        return None
    acts = list(fn()) if callable(fn) else list(fn)
    steps = int(getattr(w, "dead_time", getattr(w, "_steps", len(acts))))
    if len(acts) != steps:
        raise ValueError(f"reset_action returned {len(acts)} actions for a dead time of {steps} steps (dead_time_processor.py:13-16)")
    rows = [np.atleast_1d(np.asarray(a, dtype=float)).ravel() for a in acts]
    if any(r.shape != rows[0].shape or not np.array_equal(r, rows[0]) for r in rows[1:]):
        raise NotImplementedError("DeadTimeProcessor(reset_action=...): the accelerated path refills the queue with `steps` copies of ONE action; "
                                  f"got different actions {[r.tolist() for r in rows]}")
    return [float(x) for x in rows[0]]


class DqToAbcActionProcessor:
    """(u_d, u_q[, u_e]) actions -> abc converter actions with the Park angle advanced by (0.5 + dead time) * tau * omega * p.
    Motor types 'PMSM' (any SynchronousMotorSystem) and 'EESM'.  The reference's 'SCIM' / 'DFIM' variants read a
    'psi_angle' state that only a FluxObserver wrapper provides (observation post-processing: not on the accelerated path)."""

    _SUPPORTED = ("PMSM", "EESM")

    def __init__(self, motor_type="PMSM"):
        if motor_type not in self._SUPPORTED:
            raise NotImplementedError(f"DqToAbcActionProcessor for {motor_type!r} needs a flux observer; supported on the accelerated "
                                      f"path: {self._SUPPORTED}")
        self.motor_type = motor_type

    @classmethod
    def make(cls, motor_type, *args, **kwargs):
        return cls(motor_type)


def fold_wrappers(wrappers):
    """-> dict(action_delay=..., action_frame=...[, action_delay_reset=...]) for BatchedSCMLSystem from a reference-style wrapper tuple (innermost first).
    Accepts this module's holders and the reference's own instances (by class name)."""
    delay, frame, seen_dq, reset_row = 0, None, False, None  # frame None: leave it to the system's control_space
    for w in wrappers:
        names = {c.__name__ for c in type(w).__mro__}
        if "DeadTimeProcessor" in names:
            if seen_dq:
                raise ValueError("DeadTimeProcessor must be wrapped INSIDE the DqToAbcActionProcessor (listed before it), as the "
                                 "reference's processor expects (dq_to_abc_action_processor.py:83-86)")
            row = _reset_action_row(w)
            if delay and (row or reset_row) and row != reset_row:
                raise NotImplementedError("several DeadTimeProcessors with different reset actions are not on the accelerated path")
            reset_row = row if row is not None else reset_row
            delay += int(getattr(w, "dead_time", getattr(w, "_steps", 0)))
        elif "DqToAbcActionProcessor" in names:
            if "_DFIMDqToAbcActionProcessor" in names or getattr(w, "_angle_name", "epsilon") != "epsilon":
                raise NotImplementedError("dq processors that need a flux observer (SCIM, DFIM) are not on the accelerated path")
            frame, seen_dq = "dq_processor", True
        else:
            raise NotImplementedError(f"physical-system wrapper {type(w).__name__} is not on the accelerated path (observation "
                                      "post-processing stays on the host: wrap the n_envs=1 system with the reference's wrapper)")
    out = dict(action_delay=delay, action_frame=frame)
    if reset_row is not None and any(reset_row):
        out["action_delay_reset"] = reset_row
    return out

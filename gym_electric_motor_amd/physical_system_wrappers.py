"""Parameter-holder mirrors of the two ACTION-side physical-system wrappers that sit directly in front of `simulate()`
(SURVEY.md section 8f rank 2).  They contain no logic: `make(..., physical_system_wrappers=(...))` (or the
`action_delay=` / `action_frame=` arguments of `BatchedSCMLSystem`) folds them into the kernel's action stage, so the
fused rollout stays on the device.

    DeadTimeProcessor(steps)                 physical_system_wrappers/dead_time_processor.py:8-85
    DqToAbcActionProcessor.make(motor_type)  physical_system_wrappers/dq_to_abc_action_processor.py:9-175

As in the reference, wrappers are applied innermost first: `(DeadTimeProcessor(2), DqToAbcActionProcessor.make("PMSM"))`
delays the abc action by two steps and lets the dq processor advance its angle by 0.5 + 2 steps (lines 83-86).
Observation-side wrappers (flux observer, cos/sin, current sum, noise) are post-processing and stay on the host.
"""


class DeadTimeProcessor:
    """The converter receives the action submitted `steps` control steps earlier; every reset refills the queue with the
    zero action (the reference's default `reset_action`; custom reset actions are not on the accelerated path)."""

    def __init__(self, steps=1, reset_action=None, physical_system=None):
        if reset_action is not None:
            raise NotImplementedError("custom reset_action: only the default (zero action) is on the accelerated path")
        self._steps = int(steps)
        assert self._steps > 0, f'The number of steps has to be greater than 0. A "{steps}" has been passed.'

    @property
    def dead_time(self):
        return self._steps


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
    """-> dict(action_delay=..., action_frame=...) for BatchedSCMLSystem from a reference-style wrapper tuple (innermost first).
    Accepts this module's holders and the reference's own instances (by class name)."""
    delay, frame, seen_dq = 0, None, False  # frame None: leave it to the system's control_space
    for w in wrappers:
        names = {c.__name__ for c in type(w).__mro__}
        if "DeadTimeProcessor" in names:
            if seen_dq:
                raise ValueError("DeadTimeProcessor must be wrapped INSIDE the DqToAbcActionProcessor (listed before it), as the "
                                 "reference's processor expects (dq_to_abc_action_processor.py:83-86)")
            delay += int(getattr(w, "dead_time", getattr(w, "_steps", 0)))
        elif "DqToAbcActionProcessor" in names:
            if "_DFIMDqToAbcActionProcessor" in names or getattr(w, "_angle_name", "epsilon") != "epsilon":
                raise NotImplementedError("dq processors that need a flux observer (SCIM, DFIM) are not on the accelerated path")
            frame, seen_dq = "dq_processor", True
        else:
            raise NotImplementedError(f"physical-system wrapper {type(w).__name__} is not on the accelerated path (observation "
                                      "post-processing stays on the host: wrap the n_envs=1 system with the reference's wrapper)")
    return dict(action_delay=delay, action_frame=frame)

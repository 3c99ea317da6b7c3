"""Device-side reference generation: the batched counterpart of
`MultipleReferenceGenerator([WienerProcessReferenceGenerator(reference_state=s, ...) for s in reference_states])`
(reference_generators/wiener_process_reference_generator.py:9-49, subepisoded_reference_generator.py:11-119,
multiple_reference_generator.py:9-92) for N envs.  All generation runs in HIP kernels (csrc/gemx_refgen.hip); this class only
derives the margins the way the reference's `set_modules` does and owns the tensors.

    gen = ga.BatchedWienerProcessReferenceGenerator(reference_states=("i_sd", "i_sq"), seed=3).set_modules(env.physical_system)
    ps.set_reward(reward_weights=dict(i_sd=0.5, i_sq=0.5), referenced_states=gen.reference_names)
    gen.reset()
    refs = gen.rollout(K)                                  # [K, N, n_ref]: reference of each of the next K steps
    obs, done, reward = env.rollout(actions, references=refs)
    gen.apply_done(done)   # envs that terminated get a fresh generator state, as `if terminated: env.reset()` does

The numpy PCG64 streams of the reference cannot be reproduced on a device; the generated process is the same in distribution
(tests/test_gpu_parity.py), chunked generation equals one-shot generation bit for bit (counter-based Philox).
"""
import ctypes as C

import numpy as np

from . import _lib


class BatchedWienerProcessReferenceGenerator:
    def __init__(self, reference_states=("omega",), sigma_range=(1e-3, 1e-1), episode_lengths=(500, 2000), limit_margin=None,
                 initial_range=None, seed=0, env_base=None):
        self._reference_states = tuple(s.lower() for s in ([reference_states] if isinstance(reference_states, str) else reference_states))
        if not 1 <= len(self._reference_states) <= _lib.MAX_REF:
            raise ValueError(f"1..{_lib.MAX_REF} reference states")
        self._sigma_range = sigma_range
        self._episode_lengths = (int(episode_lengths), int(episode_lengths)) if np.ndim(episode_lengths) == 0 else tuple(int(x) for x in episode_lengths)
        self._limit_margin = limit_margin
        self._initial_range = initial_range
        self._seed = int(seed) & (2**64 - 1)
        self._env_base = None if env_base is None else int(env_base)  # None: the physical system's (a shard's generators follow its envs)
        self._handle = None

    reference_names = property(lambda self: self._ordered)

    def _margins(self, ps, name):
        """subepisoded_reference_generator.py:66-84."""
        i = ps.state_positions[name]
        low, high = ps.state_space.low[i], ps.state_space.high[i]
        lm = self._limit_margin
        if lm is None:
            f = ps.nominal_state[i] / ps.limits[i]
            return f * low, f * high
        if isinstance(lm, (float, int)):
            return lm * low, lm * high
        if isinstance(lm, tuple):
            return lm[0] * low, lm[1] * high
        raise Exception("Unknown type for the limit margin.")

    def set_modules(self, physical_system, _defer_create=False):
        ps = physical_system
        # the fused reward's reference tensor follows the state order of the physical system
        self._ordered = tuple(sorted(self._reference_states, key=lambda n: ps.state_positions[n]))
        self._n_envs = ps.n_envs
        cfg = _lib.GemxRefgenConfig()
        cfg.struct_size = C.sizeof(_lib.GemxRefgenConfig)
        cfg.n_ref = len(self._ordered)
        cfg.seed = self._seed
        cfg.env_base = self._env_base if self._env_base is not None else int(getattr(ps, "env_base", 0))
        cfg.episode_len_lo, cfg.episode_len_hi = self._episode_lengths
        for j, name in enumerate(self._ordered):
            lo, hi = self._margins(ps, name)
            cfg.margin_lo[j], cfg.margin_hi[j] = float(lo), float(hi)
            ir = self._initial_range if self._initial_range is not None else (lo, hi)  # wiener_process_reference_generator.py:25-28
            cfg.initial_lo[j], cfg.initial_hi[j] = float(ir[0]), float(ir[1])
            sr = self._sigma_range
            cfg.sigma_lo[j], cfg.sigma_hi[j] = (float(sr), float(sr)) if np.ndim(sr) == 0 else (float(sr[0]), float(sr[1]))
        self._cfg = cfg
        if _defer_create:
            return self
        import torch

        self._L = _lib.load()
        self._tdev = ps._tdev
        self._tdtype = ps._tdtype
        h = C.c_void_p()
        _lib.check(self._L.gemx_refgen_create(C.byref(cfg), self._n_envs, ps.device, _lib.F64 if self._tdtype == torch.float64 else _lib.F32, C.byref(h)))
        self._handle = h
        return self

    def _stream(self):
        import torch

        return C.c_void_p(torch.cuda.current_stream(self._tdev).cuda_stream)

    def reset(self, mask=None):
        """reference_generator.reset() for the masked envs (all by default)."""
        import torch

        m = None if mask is None else torch.as_tensor(mask).to(device=self._tdev, dtype=torch.uint8).contiguous()
        _lib.check(self._L.gemx_refgen_reset(self._handle, C.c_void_p(m.data_ptr()) if m is not None else None, self._stream()))

    def rollout(self, K, done=None, out=None):
        """References of the next K control steps, [K, N, n_ref].  done [K, N] (optional): terminations of those steps known in
        advance (e.g. a recorded rollout): generators restart after a terminating step."""
        import torch

        if out is None:
            out = torch.empty((int(K), self._n_envs, int(self._cfg.n_ref)), dtype=self._tdtype, device=self._tdev)
        d = None if done is None else done.to(device=self._tdev, dtype=torch.uint8).contiguous()
        _lib.check(self._L.gemx_refgen_rollout(self._handle, C.c_void_p(d.data_ptr()) if d is not None else None, int(K),
                                               C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def apply_done(self, done):
        """After a rollout: envs with any termination in `done` ([K, N] or [N]) restart their generators (closed-loop use)."""
        import torch

        d = done if done.dim() == 1 else done.any(dim=0)
        self.reset(mask=d.to(torch.uint8))

    def state(self):
        """(value, sigma, steps_left) per (generator, env), for tests / inspection."""
        import torch

        n = (int(self._cfg.n_ref), self._n_envs)
        v = torch.empty(n, dtype=torch.float64, device=self._tdev)
        s = torch.empty(n, dtype=torch.float64, device=self._tdev)
        l_ = torch.empty(n, dtype=torch.int32, device=self._tdev)
        _lib.check(self._L.gemx_refgen_get_state(self._handle, C.c_void_p(v.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(l_.data_ptr()), self._stream()))
        torch.cuda.current_stream(self._tdev).synchronize()
        return v, s, l_

    def close(self):
        if self._handle is not None:
            self._L.gemx_refgen_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

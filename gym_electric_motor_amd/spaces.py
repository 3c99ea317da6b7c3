"""Action / state spaces.  gymnasium's spaces are used when gymnasium is installed (so the objects are the ones
the reference's env shell expects); otherwise a minimal compatible Box / Discrete / MultiDiscrete is provided."""
import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Discrete, MultiDiscrete  # noqa: F401
except Exception:  # gymnasium is optional

    class _Space:
        def __init__(self, shape, dtype):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def __contains__(self, x):
            return self.contains(x)

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
            super().__init__(shape, dtype)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __eq__(self, other):
            return isinstance(other, Box) and self.shape == other.shape and np.allclose(self.low, other.low) and np.allclose(self.high, other.high)

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class Discrete(_Space):
        def __init__(self, n):
            self.n = int(n)
            super().__init__((), np.int64)

        def sample(self):
            return int(self._rng.integers(self.n))

        def contains(self, x):
            if isinstance(x, (int, np.integer)):
                return 0 <= int(x) < self.n
            if isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
                return 0 <= int(x) < self.n
            return False

        def __eq__(self, other):
            return isinstance(other, Discrete) and self.n == other.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiDiscrete(_Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            super().__init__(self.nvec.shape, np.int64)

        def sample(self):
            return self._rng.integers(0, self.nvec).astype(np.int64)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and np.issubdtype(x.dtype, np.integer) and bool(np.all(x >= 0) and np.all(x < self.nvec))

        def __eq__(self, other):
            return isinstance(other, MultiDiscrete) and np.array_equal(self.nvec, other.nvec)

        def __repr__(self):
            return f"MultiDiscrete({self.nvec})"

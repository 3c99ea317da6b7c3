import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
        super().__init__(shape, dtype, seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __eq__(self, other):
        return (
            isinstance(other, Box)
            and self.shape == other.shape
            and np.allclose(self.low, other.low)
            and np.allclose(self.high, other.high)
        )

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        self.n = int(n)
        self.start = int(start)
        super().__init__((), np.int64, seed)

    def sample(self):
        return int(self.start + self._rng.integers(self.n))

    def contains(self, x):
        if isinstance(x, (int, np.integer)):
            v = int(x)
        elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            v = int(x)
        else:
            return False
        return self.start <= v < self.start + self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n and self.start == other.start

    def __repr__(self):
        return f"Discrete({self.n})"


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self):
        return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.array_equal(self.nvec, other.nvec)


class Tuple(Space):
    def __init__(self, spaces, seed=None):
        self.spaces = tuple(spaces)
        super().__init__(None, None, seed)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(s.contains(p) for s, p in zip(self.spaces, x))

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)

    def __eq__(self, other):
        return isinstance(other, Tuple) and self.spaces == other.spaces

class Env:
    metadata = {"render_modes": []}
    render_mode = None
    spec = None
    action_space = None
    observation_space = None

    def reset(self, *, seed=None, options=None):
        return None

    def step(self, action):
        raise NotImplementedError

    def render(self):
        return None

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

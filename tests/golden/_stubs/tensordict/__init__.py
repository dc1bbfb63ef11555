from rl8_amd.tensordict import TensorDict  # noqa: F401

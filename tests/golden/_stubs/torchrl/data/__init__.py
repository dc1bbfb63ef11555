from rl8_amd.specs import Categorical, Composite, TensorSpec, Unbounded  # noqa: F401

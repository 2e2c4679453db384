from . import runtime_env  # noqa: F401  (process-level ROCm runtime defaults, before anything initialises HIP)

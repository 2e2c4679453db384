"""ROCm runtime settings this package prefers, applied to the PROCESS environment when the package is imported -- before the HIP
runtime initialises (it reads its flags once, at the first HIP call), after which they have no effect.

DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: the runtime dispatches the kernel nodes of a replayed hipGraph through its ordinary path
instead of from AQL packets it pre-recorded at instantiation.  The pre-recorded form saves host time per replay and costs
device time between two replays: on MI355X / ROCm 7.2 a replayed [pyramid -> PatchMatch] step is 3.5-4 us shorter without it
(0.0903 -> 0.0867 ms in one session, profiles/r04_runtime_env_probe.log), and the host is nowhere near the bottleneck of an
88-us step.  The whole GPU suite passes either way.  A value already present in the environment is left alone;
PAGK_KEEP_RUNTIME_ENV=1 switches this module off.  A C / C++ host exports the variable itself."""
import os

PREFERRED = {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}


def apply() -> dict:
    """-> what is in force for the preferred settings ({name: value}); sets the ones the environment does not define."""
    if os.environ.get("PAGK_KEEP_RUNTIME_ENV") == "1":
        return {k: os.environ.get(k) for k in PREFERRED}
    for k, v in PREFERRED.items():
        os.environ.setdefault(k, v)
    return {k: os.environ.get(k) for k in PREFERRED}


IN_FORCE = apply()

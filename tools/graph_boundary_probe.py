"""What ONE replay of a hipGraph costs between two replays (the step's fixed cost): a graph of k tiny kernels replayed back to back,
and the same tiny kernels launched directly, per replay / per launch (torch's CUDAGraph = hipGraph on ROCm)."""
import time
import torch
dev = torch.device("cuda", 0)
x = torch.zeros(64, device=dev)
s = torch.cuda.Stream(device=dev)
def timed(fn, reps=2000):
    with torch.cuda.stream(s):
        for _ in range(200):
            fn()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        s.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
with torch.cuda.stream(s):
    print(f"direct tiny kernel, back to back: {timed(lambda: x.add_(1.0)):.2f} us per launch")
    for k in (1, 2, 4):
        g = torch.cuda.CUDAGraph()
        x.add_(1.0)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(k):
                x.add_(1.0)
        print(f"graph of {k} tiny kernel(s): {timed(g.replay):.2f} us per replay")

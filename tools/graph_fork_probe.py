"""How does a hipGraph replay schedule two parallel branches? A fork / join of a chain of LONG kernels (main) and a chain of SHORT ones (side),
captured in three orders: side first, main first, interleaved. Event-timed here; run under `rocprofv3 --kernel-trace` to see the queues.
Found (round 5, profiles/r05_graph_branches.md): the branch captured SECOND starts only when the first one is (almost) through."""
import sys
import torch

order = sys.argv[1] if len(sys.argv) > 1 else "side_first"
n_main, n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 12, int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = "cuda"
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
outs = [torch.empty_like(a) for _ in range(n_main)]
small = [torch.zeros(4096, device=dev) for _ in range(n_side)]
side = torch.cuda.Stream()


def main_k(i):
    torch.matmul(a, b, out=outs[i])


def side_k(i):
    small[i].add_(1.0)


def body():
    main = torch.cuda.current_stream()
    small[0].add_(0.0)                       # the fork node
    side.wait_stream(main)
    if order == "side_first":
        with torch.cuda.stream(side):
            for i in range(n_side):
                side_k(i)
        for i in range(n_main):
            main_k(i)
    elif order == "main_first":
        for i in range(n_main):
            main_k(i)
        with torch.cuda.stream(side):
            for i in range(n_side):
                side_k(i)
    elif order == "serial":
        for i in range(n_side):
            side_k(i)
        for i in range(n_main):
            main_k(i)
    else:                                    # interleaved capture: one main kernel, then a share of the side chain
        per = -(-n_side // n_main)
        j = 0
        for i in range(n_main):
            main_k(i)
            with torch.cuda.stream(side):
                for _ in range(per):
                    if j < n_side:
                        side_k(j)
                        j += 1
    main.wait_stream(side)
    small[0].add_(0.0)                       # the join node


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"{order}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per replay ({n_main} long + {n_side} short kernels)")

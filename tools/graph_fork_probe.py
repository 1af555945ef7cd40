"""dev: does a replayed HIP graph run two forked branches concurrently?  Two chains of small-grid kernels (each far from filling
the GPU), forked (a) after a first node on the capture stream, (b) at the very beginning of the capture."""
import torch, time
dev = torch.device("cuda:0")
a = torch.randn(256, 256, device=dev); b = torch.randn(256, 256, device=dev)
N = 200


def chain(x, n):
    for _ in range(n):
        x = torch.mm(x, b) * 0.01
    return x


def run(mode):
    g = torch.cuda.CUDAGraph()
    s_main = torch.cuda.Stream(); s_side = torch.cuda.Stream()
    s_main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_main):
        chain(a, 3); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s_main):
            if mode == "serial":
                y1 = chain(a, N); y2 = chain(a, N)
            elif mode == "fork_after_first":
                p = a + 1.0                                   # a first node on the capture stream
                s_side.wait_stream(s_main)
                with torch.cuda.stream(s_side):
                    y2 = chain(p, N)
                y1 = chain(p, N)
                s_main.wait_stream(s_side)
            elif mode == "fork_after_first_main_first":
                p = a + 1.0
                s_side.wait_stream(s_main)
                y1 = chain(p, N)
                with torch.cuda.stream(s_side):
                    y2 = chain(p, N)
                s_main.wait_stream(s_side)
            elif mode == "fork_at_start":
                s_side.wait_stream(s_main)
                with torch.cuda.stream(s_side):
                    y2 = chain(a, N)
                y1 = chain(a, N)
                s_main.wait_stream(s_side)
            z = y1 + y2
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"{mode:32s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per replay", flush=True)


for m in ("serial", "fork_after_first", "fork_after_first_main_first", "fork_at_start"):
    run(m)

"""Would two half-batch pipelines on two streams fill each other's bubbles?  (round 4 experiment)
The step's kernels are short and latency-bound (12 K-tiles per GEMM, one tile per CU, epilogues that overlap nothing).  Two INDEPENDENT
replicas at B/2 replayed concurrently on two streams are the upper bound of what a micro-batched step could gain: if 2 x (B/2) concurrent is
not faster than 1 x B, splitting the batch inside one step cannot be either.
    python scripts/microbatch_probe.py [B=48]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vln_goat_amd import dp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
torch.cuda.set_device(0)


class A:
    pass


def mk(batch):
    a = A()
    a.batch, a.dtype, a.layers, a.no_autotune, a.no_graph, a.no_arena, a.in_graph_comm, a.wire = batch, 'bf16', '6,3,2', False, False, False, False, 'f32'
    cfg, model, b, gb, static = bench.build(a, 0)
    w = dp.GoatDataParallel(model, share_cfp_negatives=True)
    steps = bench.make_steps(a, model, gb, 1, w)
    return steps


def timeit(fn, n=30, warm=6):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


T = bench.TASKS
full = mk(B)
t_full = timeit(lambda i: full[T[i % 3]]())
print('1 x B=%d            : %.3f ms per step' % (B, t_full), flush=True)
h1, h2 = mk(B // 2), mk(B // 2)
t_half = timeit(lambda i: h1[T[i % 3]]())
print('1 x B=%d alone      : %.3f ms per step (x2 sequential = %.3f)' % (B // 2, t_half, 2 * t_half), flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both(i):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        h1[T[i % 3]]()
    with torch.cuda.stream(s2):
        h2[T[i % 3]]()
    cur.wait_stream(s1); cur.wait_stream(s2)


t_both = timeit(both)
print('2 x B=%d concurrent : %.3f ms per pair of steps  (%.2fx the full-batch step)' % (B // 2, t_both, t_both / t_full), flush=True)


def both_free(i):      # no join between iterations: each stream runs its own sequence
    with torch.cuda.stream(s1):
        h1[T[i % 3]]()
    with torch.cuda.stream(s2):
        h2[T[(i + 1) % 3]]()


t_free = timeit(both_free)
print('2 x B=%d free-running, different tasks: %.3f ms per pair' % (B // 2, t_free), flush=True)

#!/usr/bin/env python
"""bench.py — GOAT pre-training fwd+bwd throughput (trajectory-steps/s) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 30 --warmup 6
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): full R2R GOAT pre-train config (6 text / 3+3 cross-modal / 2 panorama
layers, vocab 50265, 208 M parameters, pretrain_src/config/r2r_GOAT_model_config.json), per-rank batch 48,
T=5 panoramas x 36 views x 768, 80-token instructions, tasks cycled mlm->sap->cfp (the shipped 1:1:1
mix_ratio), dropout 0.1 ON, random-init weights, synthetic inputs.  A "step" is one forward+backward of one
task on one batch (+ gradient all-reduce over ranks when N>1, no optimizer).  Compute dtype bf16 (f32
accumulate, f32 master weights / gradients).  On 1 GPU each task's step is captured once into a hipGraph
and replayed; dropout masks change per replay through a device-side counter.

One JSON line is printed by rank 0 (see the contract in the task statement) with `roofline` (MFMA GEMM
kernel: algorithmic FLOPs / HIP-event time per launch, measured live) and `cpu_baseline` (the CPU oracle
= a port of the reference path, timed on the host cores of this box on a bounded sample).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _goat_graph(g, **kw):
    """torch.cuda.graph through vln_goat_amd.hipops.graph: a graph whose capture forked one of the package's parallel branches is kept
    alive (ROCm 7.2 graph-destruction bug; see hipops.graph)."""
    from vln_goat_amd import hipops
    return hipops.graph(g, **kw)

TASKS = tuple(os.environ.get('GOAT_BENCH_TASKS', 'mlm,sap,cfp').split(','))   # (diagnostics: time one task alone)
# BASELINE.json configs -> (model-config overrides, synthetic batch shape, per-rank batch, task cycle, description)
WORKLOADS = {
    'config2': dict(cfg={}, batch=dict(T=5, L=80, style='survey'), per_rank=48, tasks=TASKS,
                    text='Full R2R GOAT pretrain config (pretrain_src/run_r2r_goat.sh): %(layers)s layers text/cross/pano, vocab 50265, '
                         'per-rank batch %(batch)d, T=5, 36x768 views, L=80, tasks mlm/sap/cfp 1:1:1, dropout 0.1'),
    # configs[4]: REVERIE model (object tokens after the views, object-grounding head, object-name embeddings), 160-token
    # instructions, 32 per rank (= 256 over 8), up to 20 objects per panorama, tasks of reverie_GOAT_pretrain.json
    'config5': dict(cfg=dict(name='REVERIE', obj_feat_size=768, image_prob_size=1000, obj_prob_size=1000, obj_name_vocab_size=45,
                             use_obj_name=True, pretrain_tasks=['mlm', 'mrc', 'sap', 'og', 'cfp']),
                    batch=dict(T=5, L=160, style='survey', objects=20, mrc=True, prob_size=1000), per_rank=32,
                    tasks=('mlm', 'mrc', 'sap', 'og', 'cfp'),
                    text='REVERIE GOAT pretrain config (pretrain_src/config/reverie_GOAT_pretrain.json shapes): %(layers)s layers, '
                         'vocab 50265, per-rank batch %(batch)d, T=5, 36 views + up to 20 objects x 768, L=160, tasks '
                         'mlm/mrc/sap/og/cfp cycled, dropout 0.1'),
}
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}     # dense peaks, MI355X_MICROARCH.md
# algorithmic FLOPs per trajectory-step, fwd+bwd, 1:1:1 task mix at L=80,T=5,V=36,G=22
ALGO_GFLOP_PER_TRAJ_STEP = {'mlm': 12.98, 'sap': 9.83, 'cfp': 7.6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=96, help='timed steps (default 96: a timed region of ~0.55 s, 32 of each task)')
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--batch', type=int, default=None, help='per-rank batch (default: 48 = train_batch_size of r2r_GOAT_pretrain.json; 32 for config5)')
    ap.add_argument('--workload', default='config2', choices=sorted(WORKLOADS) + ['config4'], help='BASELINE.json configuration timed as the headline '
                    '(config4: the fine-tuning iteration of map_nav_src, data-parallel over --gpus ranks)')
    ap.add_argument('--wire', default='f32', choices=['f32', 'bf16'], help='gradient exchange format at N > 1: float32 as the reference DDP, or bf16 '
                    'shards with float32 accumulation on receipt (dp.GradArena._reduce_mean_wire)')
    ap.add_argument('--in-graph-comm', action='store_true', help='capture the gradient exchange INSIDE the step hipGraph (one graph per step: '
                    'phased backward with each phase\'s all-reduce as a parallel branch); falls back to collectives between graphs on any capture error. '
                    'At N = 1 a one-rank RCCL group is created so that the capture path runs')
    ap.add_argument('--leg', default=None, choices=['config5', 'config4', 'large_batch'], help='(internal) run one extra leg alone and print its JSON')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the config 4 / config 5 / fresh-batch / optimizer legs (extra keys of the JSON line, N = 1 only)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-autotune', action='store_true', help='use the static GEMM tile/stage heuristics')
    ap.add_argument('--no-arena', action='store_true', help='per-parameter gradient tensors instead of the flat gradient arena')
    ap.add_argument('--cpu-batch', type=int, default=48, help='batch of the CPU-oracle timing (SURVEY 8d: the GPU workload, B = 48)')
    ap.add_argument('--layers', default='6,3,2', help='num_l_layers,num_top_layer,num_pano_layers')
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed launcher: start N ranks of this script under
    torch.distributed.run on this node (one process per GPU, rendezvous on 127.0.0.1 — the reference's launch shape,
    pretrain_src/utils/distributed.py:53-72) and pass their output through.  Under torchrun (RANK set) this is a no-op."""
    if args.gpus <= 1 or 'RANK' in os.environ or 'WORLD_SIZE' in os.environ:
        return None
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def launch_check(args):
    """GOAT_BENCH_LAUNCH_ONLY=1: exercise only the launch path (rank spawn, rendezvous, one all-reduce) and print the rank
    count — what tests/test_bench_launcher.py runs on CPU with the gloo backend."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    n = world
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('GOAT_DIST_BACKEND', 'gloo'))
        t = torch.ones(1)
        dist.all_reduce(t)
        n = int(t.item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'launch-check', 'n_gpus': n, 'gpus_flag': args.gpus, 'launch_only': True}))
    if world > 1:
        dist.destroy_process_group()


def setup_dist(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != max(1, args.gpus):
        print('[bench] --gpus %d but WORLD_SIZE=%d: the launcher decides; reporting n_gpus=%d' % (args.gpus, world, world), file=sys.stderr)
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1 or args.in_graph_comm:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:                      # a one-rank RCCL group: the collectives are identities, their launch / capture path is real
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
            from vln_goat_amd import dp
            dp.FORCE_COLLECTIVES[0] = True
        backend = os.environ.get('GOAT_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; 'gloo' only for single-GPU self-tests
        if args.in_graph_comm:
            # ProcessGroupNCCL recycles work events; one that was recorded inside a capture fails the watchdog's query when a later
            # eager collective reuses it (hipErrorCapturedEvent on this runtime): no recycling when collectives are captured
            os.environ.setdefault('TORCH_NCCL_CUDA_EVENT_CACHE', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return world, rank, local


def build(args, rank, workload='config2'):
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, pretrain_model, synth
    wl = WORKLOADS[workload]
    nl, nx, npano = [int(x) for x in args.layers.split(',')]
    cfg = gcfg.make_config(num_l_layers=nl, num_top_layer=nx, num_pano_layers=npano, **wl['cfg'])
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg)    # random init (reference init rule)
    model = model.cuda().train()
    vln_goat_amd.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    batch = synth.make_pretrain_batch(B=args.batch or wl['per_rank'], seed=100 + rank, **wl['batch'])
    if args.dtype == 'bf16' and not os.environ.get('GOAT_BENCH_F32_FEATURES'):
        # the pre-extracted view features live in a bf16 table (features.FeatureStore: SURVEY 8f N2) — the dtype the first Linear of the
        # bf16 model consumes; the float32 -> bf16 cast of the reference's float32 store happens once, when the store is built
        for k in ('traj_view_img_fts', 'traj_obj_img_fts'):
            if batch.get(k) is not None:
                batch[k] = batch[k].to(torch.bfloat16)
    static = None
    if workload == 'config2' and not os.environ.get('GOAT_BENCH_NO_STATIC'):
        # the device batch lives at fixed addresses with its index tensors (train_step.StaticBatch): the captured steps can be
        # fed a new host batch per replay (the fresh_batch leg); for the timed headline it is loaded once, like any batch
        from vln_goat_amd import train_step
        static = train_step.StaticBatch(cfg, batch, [t for t in wl['tasks'] if t in ('mlm', 'sap', 'cfp')])
        gb = static.gb
    else:
        from vln_goat_amd import train_step
        gb = synth.batch_to(train_step.prepare_position_features(batch), 'cuda')      # (position features cast + K-padded once, on the host)
    return cfg, model, batch, gb, static


class PhasePlan:
    """How the GOAT backward pass is cut for communication overlap at N > 1 (dp.GoatDataParallel.backward_phase):
        phase 0      heads + global / local cross-modal encoders            (their gradients are final first)
        phase 1      panorama stem  ||  text layers [cuts[0], n)             (parallel branches again)
        phase 2 ...  text layers [cuts[1], cuts[0]) ...
        last phase   text layers [0, cuts[-1]) + embeddings                  (its all-reduce is the only exposed one: kept small)
    Default cuts for n text layers: [n // 2, 1] (n >= 4), [n // 2] (n = 2, 3), none (n = 1: two phases).
    Forward hooks substitute identity views for the boundary tensors: the text-encoder output, the panorama stem's outputs
    and the output of the text layer below each cut (a layers._pair between layers: both handles are boundaries)."""

    def __init__(self, model, n_text_layers, cuts=None):
        n = n_text_layers
        if cuts is None:
            cuts = [n // 2, 1] if n >= 4 else ([n // 2] if n >= 2 else [])
        self.cuts = cuts = sorted({int(c) for c in cuts if 0 < int(c) < n}, reverse=True)
        if cuts:
            edges = [n] + cuts + [0]
            self.prefixes = []
            for k in range(len(cuts) + 1):
                pre = tuple('bert.lang_encoder.layer.%d.' % i for i in range(edges[k + 1], edges[k]))
                if k == 0:
                    pre = ('bert.img_embeddings.',) + pre
                if k == len(cuts):
                    pre = ('bert.embeddings.',) + pre
                self.prefixes.append(pre)
        else:                                # a one-layer text encoder: two phases only
            self.prefixes = [('bert.img_embeddings.', 'bert.embeddings.', 'bert.lang_encoder.')]
        self.b = {}

        def view(t):
            if isinstance(t, tuple):          # layers._pair: two autograd handles on one buffer, both are boundaries
                return tuple(view(u) for u in t)
            return t.view_as(t) if torch.is_tensor(t) else t

        def txt_hook(mod, inp, out):
            self.b['txt'] = view(out)
            return self.b['txt']

        def pano_hook(mod, inp, out):
            self.b['pano'] = tuple(view(t) for t in out)
            return self.b['pano']

        def mid_hook(j):
            def hook(mod, inp, out):
                self.b['mid', j] = view(out)
                return self.b['mid', j]
            return hook
        model.bert.lang_encoder.register_forward_hook(txt_hook)
        model.bert.img_embeddings.register_forward_hook(pano_hook)
        for j, c in enumerate(cuts):
            model.bert.lang_encoder.layer[c - 1].register_forward_hook(mid_hook(j))

    def first(self):        # boundaries of phase 0
        return [self.b['txt']] + [t for t in self.b['pano'] if torch.is_tensor(t)]

    def phases(self, wrapper, loss, grad_tensors=None):
        """generator: runs one backward phase per step, yielding its index (the caller launches the all-reduce in between)."""
        grads = None if grad_tensors is None else [grad_tensors]
        if not self.cuts:
            wrapper.backward_phase(0, [loss], grads, [self.b['txt']])
            yield 0
            wrapper.backward_phase(1, [self.b['txt']], 'grad', [])
            yield 1
            return
        seq = [self.first()]
        for j in range(len(self.cuts)):
            m = self.b['mid', j]
            seq.append(list(m) if isinstance(m, tuple) else [m])
        wrapper.backward_phase(0, [loss], grads, seq[0])
        yield 0
        for k in range(1, len(seq) + 1):
            wrapper.backward_phase(k, seq[k - 1], 'grad', seq[k] if k < len(seq) else [])
            yield k


def make_steps(args, model, gb, world, wrapper, tasks=None):
    """Returns {task: callable running one fwd+bwd(+all-reduce) step}.

    N = 1: one hipGraph per task (arena clear + forward + backward).
    N > 1: the backward pass is cut into phases (PhasePlan), each captured as its own hipGraph (one shared memory pool); the
    all-reduce of a phase's gradients is launched right after its replay and overlaps the later phases.  cfp (it contains
    the all-gather of the contrastive negatives) gets one more cut around the eager gather + InfoNCE piece."""
    from vln_goat_amd import hipops
    from vln_goat_amd.pretrain_model import cfp_losses
    TASKS = tuple(tasks) if tasks is not None else globals()['TASKS']
    hipops.manual_seed(1234)
    hipops.AUTOTUNE = not args.no_autotune     # first sight of a GEMM shape times (tile, LDS stages, split-K) candidates
    hipops.WgradQueue.FORCE_TUNE = not args.no_autotune      # (grouped weight gradients are timed even where GEMM-shape tuning is switched off below)
    hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')
    params = list(model.parameters())
    arena = [None]
    in_graph = bool(getattr(args, 'in_graph_comm', False))
    phased = world > 1 or bool(os.environ.get('GOAT_BENCH_PHASED')) or in_graph
    plan = PhasePlan(model, int(args.layers.split(',')[0])) if phased else None

    def prologue(task):
        if arena[0] is not None:
            arena[0].zero(task)                 # one fill per contiguous range of the task's non-kernel-owned slices
        else:
            for p in params:
                p.grad = None
        hipops.RngState.dev.add_(0x9E3779B1)
        wrapper.begin_step(task)                # (N > 1: the word-embedding gradient of sap / cfp steps is exchanged sparsely)

    def step_body(task):                        # single-graph step (N = 1, warm-up)
        prologue(task)
        loss = model(gb, task, compute_loss=True)
        hipops.backward_mean(loss)              # loss.mean().backward(): the mean for the log + a cached 1/n seed (no ones_like / div launches)
        return loss

    def eager_phased(task):
        prologue(task)
        loss = model(gb, task, compute_loss=True).mean()
        last = wrapper.n_phases - 1
        for k in plan.phases(wrapper, loss):
            wrapper.reduce_gradients(task, phase=k, wait=(k == last))

    use_graph = not args.no_graph and not (args.no_arena and world > 1)     # (graphs at N > 1 need the arena's static gradient storage)
    in_graph_ok = set()
    force_eager = {'cfp'} if os.environ.get('GOAT_BENCH_EAGER_CFP') else set()
    steps = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for task in TASKS:                       # warm-up: builds weight shadows, index caches, kernel attrs, GEMM tuning
            step_body(task)
            wrapper.record_usage(task)           # which parameters this task produces gradients for (static)
        if not args.no_arena:
            for p in params:
                p.grad = None
            arena[0] = wrapper.build_arena(phase_prefixes=plan.prefixes if phased else None)   # .grad = views into one HBM buffer
            if not os.environ.get('GOAT_BENCH_DENSE_EMBED'):
                wrapper.enable_sparse_embedding(model.bert.embeddings.word_embeddings.weight, [t for t in TASKS if t != 'mlm'],
                                                mixed_tasks=['mlm'] if phased and 'mlm' in TASKS else ())    # (no-ops at N = 1)
        phased = phased and arena[0] is not None
        try:
            with (hipops.Branch.like_capture() if not os.environ.get('GOAT_BENCH_NO_LIKE_CAPTURE') else contextlib.nullcontext()):   # parallel branches as in the capture: the grouped weight gradients the tuner times are the capture's groups
                for task in TASKS:
                    eager_phased(task) if phased else step_body(task)
        except Exception as e:      # never lose an N > 1 run to the overlap machinery: fall back to backward, then one all-reduce
            if not phased:
                raise
            print('[bench] phased backward failed in warm-up (%s: %s); using the plain backward + all-reduce path' % (type(e).__name__, e),
                  file=sys.stderr)
            torch.cuda.synchronize()
            phased = False
            force_eager.add('cfp')          # (its all-gather would sit inside the capture: keep the fallback path simple)
            wrapper._sparse = None
            hipops.SparseEmbedGrad.params.clear()
            hipops.SparseEmbedGrad.sink_list = None
            arena[0].no_zero.clear()
            for task in TASKS:
                step_body(task)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    def reduce_all(task):
        if world > 1:
            wrapper.reduce_gradients(task)

    def capture_phased(task, mode):
        """One graph per cut.  cfp: the forward stops at the four pooled vectors; the all-gather, the InfoNCE losses and their
        backward run eagerly between the first two graphs."""
        graphs, pool = [], None

        def cap(fn):
            g = torch.cuda.CUDAGraph()
            with _goat_graph(g, pool=pool, capture_error_mode=mode):
                out = fn()
            graphs.append(g)
            return g.pool(), out
        middle = None
        if task == 'cfp':
            def fwd():
                prologue(task)
                return torch.stack(model(gb, 'cfp', compute_loss=False), 0)          # [4, B, H] float32, autograd graph alive
            pool, packed = cap(fwd)
            dpacked = torch.zeros_like(packed)

            def middle():
                pd = packed.detach().requires_grad_(True)
                cfp_losses(pd[0], pd[1], pd[2], pd[3], model.temperature, model.cfp_gather).mean().backward()
                dpacked.copy_(pd.grad)
            middle()
            gen = plan.phases(wrapper, packed, dpacked)
        else:
            holder = {}

            def fwd():
                prologue(task)
                holder['loss'] = model(gb, task, compute_loss=True).mean()
            pool, _ = cap(fwd)
            gen = plan.phases(wrapper, holder['loss'])
        n_fwd = len(graphs)
        for _ in range(wrapper.n_phases):
            pool, _ = cap(lambda: next(gen))
        last = wrapper.n_phases - 1

        def run():
            for g in graphs[:n_fwd]:
                g.replay()
            if middle is not None:
                middle()
            for k, g in enumerate(graphs[n_fwd:]):
                g.replay()
                wrapper.reduce_gradients(task, phase=k, wait=(k == last))      # overlaps the replays that follow
        return run

    for task in TASKS:
        eager = (lambda t=task: eager_phased(t)) if phased else (lambda t=task: (step_body(t), reduce_all(t)))
        if not use_graph or task in force_eager:
            steps[task] = eager
            continue
        launch_stream = torch.cuda.current_stream()
        try:
            if world > 1:
                torch.cuda.synchronize()
                dist.barrier()
            mode = 'thread_local' if (world > 1 or in_graph) else 'global'   # thread_local: the RCCL watchdog thread may touch the HIP runtime
            if phased and in_graph:
                # ONE graph per step: prologue, forward, every backward phase, and each phase's gradient exchange forked onto the
                # communication stream right behind the phase that completes its gradients (parallel branches of the graph, joined
                # by the last exchange).  No host work between the phases; the cfp all-gather is a graph node too.
                try:
                    from vln_goat_amd import dp as _dp
                    _dp.quiesce_collectives()        # (the watchdog must have retired every eager collective before the streams capture)
                    gi = torch.cuda.CUDAGraph()
                    with _goat_graph(gi, capture_error_mode=mode):
                        eager_phased(task)
                    steps[task] = (lambda gi=gi: gi.replay())
                    in_graph_ok.add(task)
                except Exception as e:      # noqa: BLE001
                    print('[bench] in-graph communication: capture of %s failed (%s: %s); collectives between phase graphs instead'
                          % (task, type(e).__name__, e), file=sys.stderr)
                    torch.cuda.set_stream(launch_stream)
                    hipops.Branch.used, hipops.Branch._armed = set(), False
                    hipops.WgradQueue.reset()
                    torch.cuda.synchronize()
                    steps[task] = capture_phased(task, mode)
            elif phased:
                steps[task] = capture_phased(task, mode)
            else:
                ga = torch.cuda.CUDAGraph()
                with _goat_graph(ga, capture_error_mode=mode):
                    step_body(task)
                steps[task] = (lambda t=task, ga=ga: (ga.replay(), reduce_all(t)))
        except Exception as e:       # never lose the run to a capture problem: fall back to eager launches for this task
            print('[bench] hipGraph capture of %s failed (%s: %s); running it eagerly' % (task, type(e).__name__, e), file=sys.stderr)
            torch.cuda.set_stream(launch_stream)     # (a capture that dies leaves torch on its invalidated capture stream)
            hipops.Branch.used, hipops.Branch._armed = set(), False
            hipops.WgradQueue.reset()
            torch.cuda.synchronize()
            steps[task] = eager
    wrapper.launch_mode = ('in-graph' if in_graph_ok and len(in_graph_ok) == len([t for t in TASKS if t not in force_eager]) else 'phased') if phased else 'plain'
    wrapper.in_graph_tasks = sorted(in_graph_ok)
    return steps


def cpu_baseline(args, cfg):
    """The CPU oracle (a port of the reference path; validated against the imported reference in the build container) timed
    on this box's host cores as SURVEY 8d prescribes: the GPU workload itself (B = 48, T = 5, L = 80, the full model), fp32,
    dropout on, 2 warm-up steps, then the MEDIAN of >= 5 timed steps (mlm / sap / cfp cycled; more while the sample stays
    under ~30 s).  Threads: every core up to 64 — these GEMMs have M = B*80 = 3840 rows and stop scaling beyond that on a
    many-socket host; the record names the host CPU, its core count and the threads used."""
    from oracle import goat_oracle
    from vln_goat_amd import pretrain_model, synth
    avail = os.cpu_count() or 1
    ncores = min(avail, int(os.environ.get('GOAT_CPU_THREADS', '64')))
    torch.set_num_threads(ncores)
    cpu_model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    cpu_model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    sd['mlm_head.predictions.decoder.weight'] = sd['bert.embeddings.word_embeddings.weight']
    B = args.cpu_batch
    batch = synth.make_pretrain_batch(B=B, T=5, L=80, seed=7, style='survey')

    def one(task):
        for v in sd.values():
            if torch.is_tensor(v) and v.requires_grad:
                v.grad = None
        t0 = time.time()
        loss = goat_oracle.forward(cfg, sd, batch, task, compute_loss=True, training=True)
        loss.mean().backward()
        return time.time() - t0

    t_start = time.time()
    warm = [one('sap'), one('mlm')]              # 2 warm-up steps (allocator, thread pool)
    times = []
    i = 0
    while len(times) < 5 or (len(times) < 12 and time.time() - t_start < 30):
        times.append(one(TASKS[i % len(TASKS)]))
        i += 1
    med = sorted(times)[len(times) // 2]
    all_cores = None
    if avail > ncores and os.environ.get('GOAT_CPU_ALL_CORES'):
        # SURVEY 8d names the host's core count.  Opt-in, because it does not fit a default run: measured once on the GPU box
        # (profiles/round3_bench_line_default_all_cores_cpu.json: 2 x EPYC 9575F, 256 hardware threads) ONE step at 256 threads took
        # 422 s = 0.57 trajectory-steps/s against 6.1 s = 39.5 at 64 threads — the M = 3840-row GEMMs of this model do not scale
        # past one socket's physical cores, they collapse.  64 threads is therefore the honest "best CPU" figure.
        torch.set_num_threads(avail)
        t_all = one('sap')
        all_cores = {'cores': avail, 'value': round(B * 5 / t_all, 2), 's_per_step': round(t_all, 3), 'steps': 1}
        torch.set_num_threads(ncores)
    elif avail > ncores:
        all_cores = {'cores': avail, 'value': 0.57, 's_per_step': 422.2, 'measured': 'once, round 3 (profiles/round3_bench_line_default_all_cores_cpu.json); '
                     'set GOAT_CPU_ALL_CORES=1 to re-measure (about 7 minutes per step)'}
    return {'value': round(B * 5 / med, 2), 'unit': 'trajectory-steps/s', 'cores': ncores, 'kind': 'port', 'all_cores': all_cores,
            'cpu_model': cpu_model, 'cores_available': avail, 'median_s_per_step': round(med, 3),
            'sample': 'oracle/goat_oracle.py fp32 fwd+bwd, full R2R config, B=%d T=5 L=80, dropout on: median of %d timed steps '
                      '(mlm/sap/cfp cycled) after 2 warm-up steps, %d torch threads on %d available cores (%s), %.0f s of CPU work'
                      % (B, len(times), ncores, avail, cpu_model, time.time() - t_start)}


def committed_traffic(suffix):
    """fabric-side bytes per GEMM launch from the newest committed rocprofv3 --pmc passes of a workload
    (profiles/round*_pmc_gemm_traffic<suffix>.json, written by scripts/collect_round3.sh) -> (bytes, source) or (None, None)."""
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_pmc_gemm_traffic%s.json' % suffix)))
    if not tpaths:
        return None, None
    with open(tpaths[-1]) as f:
        tj = json.load(f)
    return round(tj['traffic_bytes_per_launch']), 'profiles/' + os.path.basename(tpaths[-1])


def gemm_roofline(args, model, gb, arena=None, tasks=None, cycle=None):
    """MFMA roofline of the dominant kernel family (goat_gemm_bf16 / goat_gemm_nt).

    One eager mlm+sap+cfp cycle records every GEMM launch (shape, pointers; the operand tensors are kept alive).
    The recorded launches are then captured, in order, into one hipGraph and replayed: the HIP-event time of a
    replay is pure GEMM time on the launch stream (no host gaps), with the cache behaviour of a real step
    because the sequence walks through all weights and activations.  achieved = sum of algorithmic FLOPs
    (2*M*N*K per launch) / replay time."""
    from vln_goat_amd import hipops, _lib
    hipops.PROFILE = []
    if cycle is not None:                    # (config 4: one eager rollout instead of a pre-training task cycle)
        cycle()
    with hipops.Branch.like_capture():       # parallel branches as in the captured steps: the same grouped weight-gradient launches, on their tuned plans
        for task in (tasks if tasks is not None else (TASKS if cycle is None else ())):
            if arena is not None:
                arena.zero(task)                 # same launch set as the timed steps (grouped weight gradients included)
            else:
                for p in model.parameters():
                    p.grad = None
            loss = model(gb, task, compute_loss=True)
            loss.mean().backward()
    torch.cuda.synchronize()
    recs = hipops.PROFILE
    hipops.PROFILE = None
    L = _lib.lib()

    def replay_all():
        st = torch.cuda.current_stream().cuda_stream
        for r in recs:
            name, cargs, _keep = r[4]
            rc = getattr(L, name)(st, *cargs)
            assert rc == 0, (name, rc)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        replay_all()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _goat_graph(g):
        replay_all()
    for _ in range(2):
        g.replay()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    tot_ms = e0.elapsed_time(e1) / reps
    tot_fl = sum(r[2] for r in recs)
    n = len(recs)
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    ach = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    # HBM-side traffic per launch of this kernel family: PMC counters cannot be read from inside the process, so the value
    # comes from the committed rocprofv3 --pmc passes of this same step mix (scripts/collect_profiles.sh; FETCH_SIZE and
    # WRITE_SIZE in separate passes, gfx950 corrections applied as MI355X_MICROARCH.md prescribes) — null if absent.
    traffic, tsrc = committed_traffic('')
    algo_bytes = 0.0
    for r in recs:
        if r[3][0] == 'grouped wgrad':
            algo_bytes += r[3][2]
            continue
        M_, N_, K_, epi_, split_ = r[3][:5]
        f32out = r[4][1][2] == 0 if r[4][0] == 'goat_gemm_bf16' else False
        algo_bytes += (M_ * K_ + N_ * K_) * 2 + M_ * N_ * (4 if f32out else 2) * (2 if epi_ in (1, 2) else 1) \
            + (M_ * N_ * 2 if epi_ in (3, 4) else 0)
    return {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'traffic': traffic, 'traffic_unit': 'bytes/launch (L2-miss reads x2-corrected + writes; rocprofv3 --pmc)',
            'traffic_source': tsrc, 'algorithmic_bytes_per_launch': round(algo_bytes / max(n, 1)), 'kernel': 'gemm2_kernel / pp_kernel / gemm2_group_kernel / pp_group_kernel / pp_group_sk_kernel (goat_gemm_bf16, goat_wgrad_grouped[_balanced]) + gemm_nt_kernel', 'launches_per_cycle': n,
            'avg_launch_us': round(tot_ms * 1e3 / max(n, 1), 2),
            'algorithmic_gflop_per_launch': round(tot_fl / max(n, 1) / 1e9, 3),
            'gemm_ms_per_cycle': round(tot_ms, 3),
            'method': 'HIP events around a hipGraph replay of the %d recorded GEMM launches of one %s' % (
                n, 'rollout' if cycle is not None else '+'.join(tasks if tasks is not None else TASKS) + ' cycle')}


def measure_pretrain(args, world, rank, workload, n_steps, n_warmup):
    """Builds the model + batch of `workload`, captures its steps and times n_steps of them (the bench contract: warm-up,
    barrier + synchronize on both sides, max over ranks).  -> dict with the timing and the live objects."""
    from vln_goat_amd import dp, synth
    wl = WORKLOADS[workload]
    tasks = tuple(wl['tasks'])
    cfg, model, batch, gb, static = build(args, rank, workload)
    wrapper = dp.GoatDataParallel(model, share_cfp_negatives=True, wire_dtype=torch.bfloat16 if args.wire == 'bf16' else None)
    wrapper.sparse_uniform_rows = True       # every rank's synthetic batch has B*L token rows (no count exchange / host sync)
    steps = make_steps(args, model, gb, world, wrapper, tasks)

    def run(i):
        steps[tasks[i % len(tasks)]]()          # fwd + bwd (+ gradient all-reduce at N > 1)

    dt = timed(run, n_steps, n_warmup, world)
    return {'dt': dt, 'n_traj': synth.n_traj_steps(batch), 'cfg': cfg, 'model': model, 'batch': batch, 'gb': gb,
            'wrapper': wrapper, 'steps': steps, 'tasks': tasks, 'run': run, 'static': static}


def timed(run, n_steps, n_warmup, world):
    for i in range(n_warmup):
        run(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_steps):
        run(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def leg_process(args, name):
    """config 4 / config 5 legs run in a child process (own model, own graphs): whatever happens there, the headline line of
    this process is printed.  -> the child's JSON object, or {'error': ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--leg', name, '--steps', str(args.steps), '--dtype', args.dtype, '--layers', args.layers]
    cmd += ['--no-graph'] if args.no_graph else []
    cmd += ['--no-roofline'] if args.no_roofline else []
    cmd += ['--no-autotune'] if args.no_autotune else []
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return {'error': 'timed out after 900 s'}
    lines = [x for x in r.stdout.splitlines() if x.startswith('{')]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stderr[-2000:])
        return {'error': 'exit code %d: %s' % (r.returncode, r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else '')}
    return json.loads(lines[-1])


def leg(name, fn):
    """extra legs never cost the headline: a failure is reported in place of the numbers."""
    try:
        return fn()
    except Exception as e:      # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        torch.cuda.synchronize()
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}


def fresh_batch_leg(args, m):
    """The captured steps fed with a NEW host batch per step.  Inside the timed region, per
    step: the batch's index tensors are rebuilt on the host from its viewpoint-id strings (graphmap.py), written into the
    batch's pinned buffer, the whole batch (27 MB: features, ids, labels, map tensors, indices) goes H2D in one copy on a side
    stream, is swapped into the fixed-address batch with one D2D copy, the masks are recomputed, and the task's hipGraph is
    replayed.  The host batches (3, cycled) were collated into pinned buffers beforehand, as a loader's workers would leave them.
    Second figure: RAGGED batches (T ~ U{1..7}, L ~ U{20..80}: every step another shape) through the eager path — H2D of the
    batch, lazy index build, eager launches (static GEMM heuristics: no first-sight tuning per shape)."""
    from vln_goat_amd import hipops, synth
    sb, tasks, steps, wl = m['static'], m['tasks'], m['steps'], WORKLOADS['config2']
    if sb is None:
        return {'error': 'no static batch'}
    B = args.batch or wl['per_rank']
    K = 4
    hosts = [synth.make_pretrain_batch(B=B, seed=500 + k, **wl['batch']) for k in range(K)]
    for h in hosts:
        h['traj_view_img_fts'] = h['traj_view_img_fts'].to(sb.gb['traj_view_img_fts'].dtype)      # (bf16 feature store rows)
    bufs = [sb.pack(h) for h in hosts]
    done = [None] * K

    def prefetch(i):                                 # host index build + H2D of batch i (side stream), one step ahead of its use
        k = i % K
        if done[k] is not None:
            done[k].synchronize()                    # this buffer's previous H2D has been executed
        sb.pack(hosts[k], bufs[k], tensors=False)    # host: index build from the id strings
        done[k] = sb.stage(bufs[k])                  # side stream: H2D (waits until the previous commit has read the staging buffer)
    state = {'next': None}

    def run(i):
        if state['next'] != i:
            prefetch(i)
        sb.commit()                                  # compute stream: swap batch i in + masks
        steps[tasks[i % len(tasks)]]()
        prefetch(i + 1)                              # overlaps the step just launched
        state['next'] = i + 1
    n = max(6, min(args.steps, 30))
    state['next'] = None
    dt = timed(run, n, 3, 1)
    sb.commit()                                      # (drain the batch staged by the last step)
    out = {'ms_per_step': round(dt / n * 1e3, 3), 'value': round(m['n_traj'] * n / dt, 1), 'unit': 'trajectory-steps/s', 'steps': n,
           'h2d_bytes_per_step': sb.nbytes,
           'what': 'new host batch per step (fixed shape B=%d T=5 L=80): host index build + one pinned H2D (side stream) + one D2D swap '
                   '+ mask refresh + hipGraph replay, all inside the timed region' % B}
    # ragged, eager
    model, arena = m['model'], m['wrapper'].arena
    rs = __import__('numpy').random.RandomState(7)
    ragged = [synth.make_pretrain_batch(B=B, T=rs.randint(1, 8, B).tolist(), L=rs.randint(20, 81, B).tolist(), seed=600 + k, style='survey')
              for k in range(6)]
    auto, hipops.AUTOTUNE = hipops.AUTOTUNE, False

    def eager(i):
        task = tasks[i % len(tasks)]
        gb = synth.batch_to(ragged[i % len(ragged)], 'cuda')
        arena.zero(task)
        hipops.RngState.dev.add_(0x9E3779B1)
        model(gb, task, compute_loss=True).mean().backward()
    try:
        n2 = 12
        dt2 = timed(eager, n2, 6, 1)
    finally:
        hipops.AUTOTUNE = auto
    traj = sum(synth.n_traj_steps(ragged[i % len(ragged)]) for i in range(n2))
    out['ragged_eager'] = {'ms_per_step': round(dt2 / n2 * 1e3, 3), 'value': round(traj / dt2, 1), 'unit': 'trajectory-steps/s',
                           'steps': n2, 'what': 'B=%d, T ~ U{1..7}, L ~ U{20..80}, a new shape every step: pageable H2D + lazy index '
                                                'build + eager launches (host-bound)' % B}
    try:
        out['ragged_bucketed'] = ragged_bucket_leg(args, m, B)
    except Exception as e:      # noqa: BLE001  (never lose the headline to an auxiliary leg)
        out['ragged_bucketed'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def ragged_bucket_leg(args, m, B):
    """RAGGED batches through captured steps: trajectory lengths T ~ U{3..6} and text lengths L ~ U{40..80} per sample —
    every batch another shape, as P/data/tasks.py's collate functions produce them — padded into a small set of SHAPE BUCKETS
    (train_step.StaticBatch(bucket=...): text 80, panoramas 224 / 256 / 288, map 32) with bf16 features as a
    features.FeatureStore hands them out.  Per step, inside the timed region: host index build + padding into the pinned buffer
    of the batch's bucket, one H2D, one D2D swap, mask refresh, replay of that bucket's hipGraph."""
    import numpy as np
    from vln_goat_amd import hipops, synth, train_step
    model, arena, tasks, cfg = m['model'], m['wrapper'].arena, m['tasks'], m['cfg']
    rs = np.random.RandomState(11)
    wl = WORKLOADS['config2']
    feat_dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32

    def mk(k):
        b = synth.make_pretrain_batch(B=B, T=rs.randint(3, 7, B).tolist(), L=rs.randint(40, 81, B).tolist(), seed=700 + k, style='survey')
        b['traj_view_img_fts'] = b['traj_view_img_fts'].to(feat_dt)
        return b
    hosts = [mk(k) for k in range(8)]
    n_buckets = (224, 256, 288)
    G = 32
    if max(h['gmap_step_ids'].shape[1] for h in hosts) > G or max(h['traj_view_img_fts'].shape[0] for h in hosts) > n_buckets[-1]:
        raise ValueError('a synthetic batch exceeds the largest bucket')
    which = [next(n for n in n_buckets if n >= h['traj_view_img_fts'].shape[0]) for h in hosts]
    sbs, graphs = {}, {}
    side = torch.cuda.Stream()
    for nb in sorted(set(which)):
        first = hosts[which.index(nb)]
        sb = train_step.StaticBatch(cfg, first, tasks, bucket=dict(L=wl['batch']['L'], N=nb, G=G))
        sbs[nb] = sb

        def body(task, gb=sb.gb):
            arena.zero(task)
            hipops.RngState.dev.add_(0x9E3779B1)
            model(gb, task, compute_loss=True).mean().backward()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), hipops.Branch.like_capture():
            for _ in range(2):
                for t in tasks:
                    body(t)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for t in tasks:
            g = torch.cuda.CUDAGraph()
            with _goat_graph(g):
                body(t)
            graphs[(nb, t)] = g
    bufs = [sbs[w].pack(h) for w, h in zip(which, hosts)]
    K = len(hosts)
    done = [None] * K
    state = {'next': None}

    host_t = [0.0, 0, 0.0]

    def prefetch(i):
        k = i % K
        if done[k] is not None:
            done[k].synchronize()
        t0 = time.perf_counter()
        sbs[which[k]].pack(hosts[k], bufs[k])        # host: padding + index build of this ragged batch into its bucket's layout
        host_t[0] += time.perf_counter() - t0
        host_t[1] += 1
        host_t[2] = max(host_t[2], time.perf_counter() - t0)
        done[k] = sbs[which[k]].stage(bufs[k])

    opt = [None]

    def run(i):
        k = i % K
        if state['next'] != i:
            prefetch(i)
        sbs[which[k]].commit()
        graphs[(which[k], tasks[i % len(tasks)])].replay()
        if opt[0] is not None:
            opt[0].step(tasks[i % len(tasks)], max_norm=5.0)
        prefetch(i + 1)
        state['next'] = i + 1
    n = 36          # (a single host stall — 40-90 ms ones were seen on the pool's boxes — weighs 1-2 ms in the mean of 36 steps; the
    dt = timed(run, n, 4, 1)      #  worst pack time of the run is reported next to the mean)
    for sb in sbs.values():
        if sb._pending is not None:
            sb.commit()
    # ---- the whole training loop of P/train_r2r_goat.py:301-366: the same ragged-bucketed input path, the captured
    # fwd+bwd AND the fused clip + AdamW update, in one timed loop; next to it the device side alone (replay + update, no new data)
    train = None
    if not os.environ.get('GOAT_BENCH_NO_TRAIN_LOOP'):
        from vln_goat_amd import optim
        opt[0] = optim.FusedAdamW(model.named_parameters(), arena, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01)
        state['next'] = None
        host_loop = list(host_t)
        dt_t = timed(run, n, 4, 1)
        for sb in sbs.values():
            if sb._pending is not None:
                sb.commit()
        nb1 = which[0]
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e2.record()
        for r in range(6):
            graphs[(nb1, tasks[r % len(tasks)])].replay()
            opt[0].step(tasks[r % len(tasks)], max_norm=5.0)
        e3.record()
        torch.cuda.synchronize()
        dev_ms = e2.elapsed_time(e3) / 6
        traj_t = sum(synth.n_traj_steps(hosts[i % K]) for i in range(n))
        train = {'ms_per_step': round(dt_t / n * 1e3, 3), 'value': round(traj_t / dt_t, 1), 'unit': 'trajectory-steps/s', 'steps': n,
                 'replay_plus_optimizer_ms_per_step': round(dev_ms, 3), 'ratio_to_replay_plus_optimizer': round(dt_t / n * 1e3 / dev_ms, 3),
                 'host_pack_ms': round((host_t[0] - host_loop[0]) / max(1, host_t[1] - host_loop[1]) * 1e3, 2),
                 'what': 'the loop of pretrain_src/train_r2r_goat.py:301-366 per update: new ragged host batch (host padding + index build into '
                         'its shape bucket, one pinned H2D on a side stream, one 27 MB D2D swap, mask refresh) + hipGraph replay of fwd+bwd + '
                         'goat_grad_sqnorm + goat_adamw_step (clip 5.0), all inside the timed region; ratio = this / (replay + update with no new data)'}
        opt[0] = None
    # the device side alone: replays of one bucket's three graphs (no new data)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nb0 = which[0]
    torch.cuda.synchronize()
    e0.record()
    for r in range(6):
        graphs[(nb0, tasks[r % len(tasks)])].replay()
    e1.record()
    torch.cuda.synchronize()
    replay_ms = e0.elapsed_time(e1) / 6
    traj = sum(synth.n_traj_steps(hosts[i % K]) for i in range(n))
    real = float(np.mean([h['traj_view_img_fts'].shape[0] for h in hosts]))
    return {'ms_per_step': round(dt / n * 1e3, 3), 'value': round(traj / dt, 1), 'unit': 'trajectory-steps/s', 'steps': n,
            'buckets': {'L': wl['batch']['L'], 'N': list(sorted(set(which))), 'G': G}, 'mean_panoramas_per_batch': round(real, 1),
            'feature_dtype': str(feat_dt).replace('torch.', ''), 'host_pack_ms': round(host_t[0] / max(1, host_t[1]) * 1e3, 2), 'host_pack_ms_max': round(host_t[2] * 1e3, 2),
            'replay_only_ms_per_step': round(replay_ms, 3), 'train_loop': train,
            'what': 'B=%d, T ~ U{3..6}, L ~ U{40..80}: a new shape every step, padded into %d shape buckets; per step host padding + index '
                    'build, one pinned H2D, D2D swap, mask refresh, hipGraph replay of the bucket (all inside the timed region)' % (B, len(set(which)))}


def optimizer_leg(args, m):
    """fwd + bwd + the fused clip + AdamW update (P/train_r2r_goat.py:349-366 per update) on the headline
    workload: the captured step, then optim.FusedAdamW.step(task) on the gradient arena (two kernels, no host sync)."""
    from vln_goat_amd import optim
    opt = optim.FusedAdamW(m['model'].named_parameters(), m['wrapper'].arena, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01)
    tasks, steps = m['tasks'], m['steps']

    def run(i):
        t = tasks[i % len(tasks)]
        steps[t]()
        opt.step(t, max_norm=5.0)
    n = max(6, min(args.steps, 30))
    dt = timed(run, n, 3, 1)
    return {'ms_per_step': round(dt / n * 1e3, 3), 'value': round(m['n_traj'] * n / dt, 1), 'unit': 'trajectory-steps/s', 'steps': n,
            'what': 'hipGraph replay of fwd+bwd, then goat_grad_sqnorm + goat_adamw_step (clip 5.0, lr 5e-5, betas 0.9/0.98, decay 0.01; '
                    'bf16 weight shadows refreshed by the update kernel)'}


def config5_leg(args):
    n = max(10, min(args.steps, 30))
    m = measure_pretrain(args, 1, 0, 'config5', n, 5)
    out = {'value': round(m['n_traj'] * n / m['dt'], 1), 'unit': 'trajectory-steps/s', 'ms_per_step': round(m['dt'] / n * 1e3, 3),
           'steps': n, 'samples_per_s': round(m['n_traj'] * n / m['dt'] / 5.0, 1),
           'workload': WORKLOADS['config5']['text'] % {'layers': args.layers, 'batch': args.batch or 32} + ', fwd+bwd, hipGraph replay'}
    if not args.no_roofline:
        r = gemm_roofline(args, m['model'], m['gb'], m['wrapper'].arena, tasks=m['tasks'])
        out['roofline'] = {k: r[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'launches_per_cycle', 'avg_launch_us',
                                             'algorithmic_gflop_per_launch', 'algorithmic_bytes_per_launch', 'gemm_ms_per_cycle', 'method')}
        out['roofline']['traffic'], out['roofline']['traffic_source'] = committed_traffic('_config5')
    m.clear()
    return out


def large_batch_leg(args):
    """The headline workload (configs[1]: same model, tasks and shapes) at per-rank batch 192 and 256: where the MFMA fraction of the
    GEMM family and of the whole step are no longer set by B = 48's 12-K-tile, one-tile-per-CU launches (2.4 TFLOP per
    step at B = 48).  Each batch size: captured steps timed as the headline is, its own roofline block, peak HBM footprint."""
    import gc
    out = {'what': 'BASELINE.json configs[1] at larger per-rank batches (same model / tasks / shapes, hipGraph replay, dropout on, fwd+bwd); '
                   'step_mfma_frac = algorithmic FLOPs of the step / time / dense bf16 MFMA peak'}
    algo = sum(ALGO_GFLOP_PER_TRAJ_STEP.values()) / 3.0
    for B in [int(x) for x in os.environ.get('GOAT_BENCH_LARGE_BATCHES', '192,256').split(',')]:
        args.batch = B
        torch.cuda.reset_peak_memory_stats()
        n = max(9, min(args.steps, 15)) // 3 * 3
        m = measure_pretrain(args, 1, 0, 'config2', n, 3)
        value = m['n_traj'] * n / m['dt']
        rec = {'per_rank_batch': B, 'value': round(value, 1), 'unit': 'trajectory-steps/s', 'ms_per_step': round(m['dt'] / n * 1e3, 3), 'steps': n,
               'step_mfma_frac': round(value * algo * 1e9 / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12), 4)}
        if not args.no_roofline:
            r = gemm_roofline(args, m['model'], m['gb'], m['wrapper'].arena, tasks=m['tasks'])
            rec['roofline'] = {k: r[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'launches_per_cycle', 'avg_launch_us',
                                                 'algorithmic_gflop_per_launch', 'algorithmic_bytes_per_launch', 'gemm_ms_per_cycle', 'method')}
            rec['roofline']['traffic'] = None
        rec['peak_hbm_gib'] = round(torch.cuda.max_memory_allocated() / 2**30, 2)
        out['B%d' % B] = rec
        m.clear()
        del m
        gc.collect()
        torch.cuda.empty_cache()
    return out


_T0 = [None]


def _progress(msg):
    """GOAT_BENCH_PROGRESS=1: timestamped stage marks on stderr (where a leg's wall time goes; the JSON line on stdout is untouched)"""
    if os.environ.get('GOAT_BENCH_PROGRESS'):
        import time as _t
        _T0[0] = _T0[0] or _t.perf_counter()
        print('[bench +%.1fs] %s' % (_t.perf_counter() - _T0[0], msg), file=sys.stderr, flush=True)


def config4_leg(args, rank=0, world=1):
    """BASELINE.json configs[3] per rank: the fine-tuning model's calls of one rollout (text once, then panorama + navigation
    per step with the [MEM] token carried: back-propagation through time) with BACL + FACL on, at the shapes of
    M/scripts/run_r2r_goat.sh (batch 12 per rank, max_instr_len 200, dictionaries 35/39/50/24, G = 60 map nodes): forward,
    imitation loss, backward.  A trajectory-step = one panorama of one sample."""
    import vln_goat_amd
    from types import SimpleNamespace
    from vln_goat_amd import nav_model, synth, hipops
    a = SimpleNamespace(num_l_layers=6, num_x_layers=3, num_pano_layers=2, dropout=0.1, feat_dropout=0.5, vocab_size=50265,
                        do_back_img=True, do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True,
                        do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door', mode='train')
    torch.manual_seed(0)
    model = nav_model.GlocalTextPathNavCMT(nav_model.nav_config_from_args(a)).cuda().train()
    vln_goat_amd.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    # T = 6: the length of an R2R ground-truth path (5-7 viewpoints; the teacher-forced rollout of M/r2r/agent.py stops at its end; the
    # reference caps rollouts at max_action_len 15, scripts/run_r2r_goat.sh:35 — GOAT_NAV_T=15 runs that as a stress case).  Round 3 timed T = 3.
    B, T = int(os.environ.get('GOAT_NAV_B', '12')), int(os.environ.get('GOAT_NAV_T', '6'))
    ep = synth.make_nav_episode(B=B, L=200, n_steps=T, seed=21 + rank, vocab_size=50265, extra_nodes=54 - T)      # G = 60 map nodes at the last step; every rank rolls out its own shard
    mv = lambda x: x.cuda() if torch.is_tensor(x) else x
    for st in ep['steps']:            # the agent's collate: logit-fusion matrix of the step from the id strings (host)
        st['nav_fusion'] = nav_model.nav_fusion_matrix(st['vp_cand_vpids'], st['gmap_vpids'], st['gmap_visited_masks'],
                                                       st['gmap_step_ids'].shape[1], st['vp_masks'].shape[1])
    from vln_goat_amd import train_step
    ep['steps'] = [train_step.prepare_position_features(st) for st in ep['steps']]      # (cast + K-pad of the 7- / 14-wide features: once, on the host)
    ep = {k: ([{kk: mv(vv) for kk, vv in st.items()} for st in v] if k == 'steps' else mv(v)) for k, v in ep.items()}
    params = list(model.parameters())
    hipops.manual_seed(4321)
    hipops.AUTOTUNE = not args.no_autotune     # first sight of a GEMM shape times (tile, LDS stages, split-K) candidates
    hipops.WgradQueue.FORCE_TUNE = not args.no_autotune      # (grouped weight gradients are timed even where GEMM-shape tuning is switched off below)
    hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')       # per-replay dropout counter

    arena = [None]
    from vln_goat_amd import dp
    # M/r2r/agent_base.py:100-102: vln_bert and critic both under data parallelism (rank 0's weights broadcast at construction)
    critic = nav_model.Critic(a).cuda() if hasattr(nav_model, 'Critic') else None
    wrapper, wcritic = dp.wrap_finetune_models(model, critic, wire_dtype=torch.bfloat16 if getattr(args, 'wire', 'f32') == 'bf16' else None)
    in_graph = bool(getattr(args, 'in_graph_comm', False))

    def episode():
        if arena[0] is not None:
            arena[0].zero('nav')             # gradient arena: weight gradients of the rollout's Linears are grouped (goat_wgrad_grouped)
        else:
            for p in params:
                p.grad = None
        hipops.RngState.dev.add_(0x9E3779B1)
        # (the instruction's K|V projections of the six cross-modal layers once per episode instead of once per step: identical
        #  outputs — tests/test_nav_parity_gpu.py holds both forms to the reference goldens)
        loss, _ = synth.run_nav_episode(lambda m, b: model(m, b), ep, device='cuda', hoist_text_kv=not os.environ.get('GOAT_NAV_NO_HOIST'),
                                        hoist_pano=not os.environ.get('GOAT_NAV_NO_HOIST') and not os.environ.get('GOAT_NAV_NO_PANO_HOIST'))
        loss.backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            episode()
        if not args.no_arena:
            wrapper.record_usage('nav')
            for p in params:
                p.grad = None
            arena[0] = wrapper.build_arena()
        with hipops.Branch.like_capture():
            for _ in range(2):
                episode()
                dp.reduce_finetune_gradients((wrapper, wcritic))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    exchange = lambda: dp.reduce_finetune_gradients((wrapper, wcritic))      # no-op at N = 1 (unless --in-graph-comm forces the one-rank group)
    run, launch = (lambda i: (episode(), exchange())), 'eager'
    if not args.no_graph:
        launch_stream = torch.cuda.current_stream()
        mode = 'thread_local' if (world > 1 or in_graph) else 'global'

        def reset_capture(e, what):
            print('[bench] hipGraph capture of %s failed (%s: %s)' % (what, type(e).__name__, e), file=sys.stderr)
            torch.cuda.set_stream(launch_stream)
            hipops.Branch.used, hipops.Branch._armed = set(), False
            hipops.WgradQueue.reset()
            torch.cuda.synchronize()
        done = False
        if in_graph and arena[0] is not None:
            try:
                dp.quiesce_collectives()
                g = torch.cuda.CUDAGraph()
                with _goat_graph(g, capture_error_mode=mode):
                    episode()
                    exchange()
                run, launch, done = (lambda i: g.replay()), 'hipGraph replay, gradient exchange inside the graph', True
            except Exception as e:      # noqa: BLE001
                reset_capture(e, 'the navigation episode with its gradient exchange')
        if not done:
            try:
                if world > 1:
                    torch.cuda.synchronize()
                    dist.barrier()
                g = torch.cuda.CUDAGraph()
                with _goat_graph(g, capture_error_mode=mode):
                    episode()
                run, launch = (lambda i: (g.replay(), exchange())), 'hipGraph replay' + (', then the gradient exchange' if world > 1 else '')
            except Exception as e:      # noqa: BLE001
                reset_capture(e, 'the navigation episode')
    _progress('config4: episode captured, timing')
    n = max(6, min(args.steps, 20)) if world == 1 and args.leg else args.steps
    dt = timed(run, n, 2 if world == 1 and args.leg else args.warmup, world)
    dp_diag = None
    if world > 1 or in_graph:
        # the diagnostics block of the pre-training workload, for the fine-tuning iteration: ranks seen, compute-only time, exchange alone
        try:
            info = [None] * world
            if world > 1:
                dist.all_gather_object(info, {'rank': rank, 'device': torch.cuda.current_device(), 'name': torch.cuda.get_device_name(), 'pid': os.getpid()})
            nn_ = max(6, min(n, 12))
            dt_c = timed((lambda i: g.replay()) if (not args.no_graph and not in_graph) else (lambda i: episode()), nn_, 2, world)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                exchange()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            nbytes = int(sum((b_ - a_) * 4 for a_, b_ in arena[0].ranges('nav', None, frozenset()))) if arena[0] is not None else None
            dp_diag = {'ranks_seen_by_rccl': dist.get_world_size(), 'backend': dist.get_backend(), 'ranks': info if world > 1 else None,
                       'launch_mode': launch, 'wire': getattr(args, 'wire', 'f32'), 'ms_per_episode': round(dt / n * 1e3, 3),
                       'compute_only_ms_per_episode': round(dt_c / nn_ * 1e3, 3) if not in_graph else None,
                       'exposed_comm_ms_per_episode': round(dt / n * 1e3 - dt_c / nn_ * 1e3, 3) if not in_graph else None,
                       'exchange_alone': {'ms': round(ms, 3), 'bytes': nbytes,
                                          'bus_GBps': round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1) if nbytes and world > 1 else None}}
        except Exception as e:      # noqa: BLE001
            dp_diag = {'error': '%s: %s' % (type(e).__name__, e)}
    _progress('config4: timed; roofline leg')
    roof = None
    if not args.no_roofline and world == 1:
        r = gemm_roofline(args, model, None, None, cycle=episode)
        roof = {k: r[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'launches_per_cycle', 'avg_launch_us',
                                  'algorithmic_gflop_per_launch', 'algorithmic_bytes_per_launch', 'gemm_ms_per_cycle', 'method')}
        roof['traffic'], roof['traffic_source'] = committed_traffic('_config4')
    _progress('config4: navigator leg')
    nav = None
    if not args.no_graph and not os.environ.get('GOAT_BENCH_NO_NAVIGATOR') and world == 1 and not in_graph:
        try:
            nav = navigator_leg(args, model, ep, arena[0], B, T, dt / n)
        except Exception as e:      # noqa: BLE001
            nav = {'error': '%s: %s' % (type(e).__name__, e)}
        if not os.environ.get('GOAT_BENCH_NO_REVERIE_NAV'):
            _progress('config4: REVERIE navigator leg')
            try:
                nav['reverie'] = reverie_navigator_leg(args, ep, B, T)
            except Exception as e:      # noqa: BLE001
                nav['reverie'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    return {'value': round(B * T * world * n / dt, 1), 'unit': 'trajectory-steps/s', 'ms_per_episode': round(dt / n * 1e3, 3), 'episodes': n,
            'launch': launch, 'roofline': roof, 'navigator': nav, 'dp': dp_diag, 'per_rank_batch': B, 'steps_per_episode': T,
            'workload': 'map_nav_src fine-tune model calls of one rollout (run_r2r_goat.sh shapes): 6,3,2 layers, batch 12, L=200, %d steps x ' % T +
                        '(panorama 36x768 + navigation, G=60), BACL+FACL on (type_2 / type_1 / door), dictionaries 35/39/50/24, dropout '
                        '0.1 / feat 0.5, BPTT through the [MEM] token, fwd+bwd, synthetic per-step inputs (no simulator)'}


def dp_diagnostics(args, m, world, rank, dt_step):
    """N > 1 only — what makes the first multi-GPU run diagnosable from its one JSON line: the ranks RCCL actually sees (count,
    backend, device of every rank), the same steps with the gradient exchange switched OFF (compute-only time per step; the
    difference to the timed step is the EXPOSED communication), and the gradient all-reduce of each task run alone (its un-hidden
    cost, bytes and bus bandwidth).  Collective on every rank; rank 0 reports."""
    wrapper, tasks, steps = m['wrapper'], m['tasks'], m['steps']
    info = [None] * world
    dist.all_gather_object(info, {'rank': rank, 'device': torch.cuda.current_device(), 'name': torch.cuda.get_device_name(),
                                  'pid': os.getpid()})
    real = wrapper.reduce_gradients
    wrapper.reduce_gradients = lambda *a, **k: None
    try:
        n = max(6, min(args.steps, 18))
        dt_c = timed(lambda i: steps[tasks[i % len(tasks)]](), n, 3, world)
    finally:
        wrapper.reduce_gradients = real
    alone = {}
    for t in tasks:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        nph = getattr(wrapper, 'n_phases', None)
        for _ in range(3):
            if nph:                              # the phased exchange of the timed step, without the backward phases between its parts
                for k in range(nph):
                    wrapper.reduce_gradients(t, phase=k, wait=(k == nph - 1))
            else:
                wrapper.reduce_gradients(t)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        nbytes = None
        if wrapper.arena is not None:
            try:
                nbytes = int(sum((b - a) * 4 for a, b in wrapper.arena.ranges(t, None, frozenset())))
            except Exception:      # noqa: BLE001
                nbytes = None
        alone[t] = {'ms': round(ms, 3), 'bytes': nbytes,
                    'bus_GBps': round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1) if nbytes else None}
    step_ms, comp_ms = dt_step / args.steps * 1e3, dt_c / n * 1e3
    return {'ranks_seen_by_rccl': dist.get_world_size(), 'backend': dist.get_backend(), 'ranks': info, 'launch_mode': wrapper.launch_mode,
            'n_phases': getattr(wrapper, 'n_phases', None), 'ms_per_step': round(step_ms, 3), 'compute_only_ms_per_step': round(comp_ms, 3),
            'exposed_comm_ms_per_step': round(step_ms - comp_ms, 3), 'allreduce_alone': alone}


def dagger_iteration(model, te, bufs, batches, extras, arena, sim, store, B, T, ml_weight=0.2, max_action_len=15):
    """The reference's training iteration (train_alg=dagger, M/r2r/agent.py:414-445 with scripts/run_r2r_goat.sh:35,41-42): a TEACHER rollout
    weighted ml_weight = 0.2 and a SAMPLE rollout (the policy's own sampled actions, up to max_action_len = 15 steps, loss against the
    teacher action of every visited state), the two losses summed, ONE backward.  Here: the sampled rollout is host-driven (the next
    observation depends on the sampled action: one B-element device -> host copy per step, eager launches; rollout.NavRollout) and
    back-propagates first; the teacher rollout is the captured episode graph, captured in ACCUMULATE form (no arena clear: every gradient
    slice was already written in this step, so every kernel of the replay adds) with its loss scaled by ml_weight — together
    d(L_sample + 0.2 L_teacher), what the reference's single backward produces."""
    from vln_goat_amd import hipops, rollout
    call = lambda mode, batch: model(mode, batch)
    ro = rollout.NavRollout(call, sim, store, max_action_len=max_action_len, pano_width=38, gmap_buckets=[64, 96, 128])

    def sample_part(i):
        arena.zero('nav')
        hipops.RngState.dev.add_(0x9E3779B1)
        loss, _ = ro.run(batches[i % len(batches)], feedback='sample', extras=extras, train_ml=1.0)
        loss.backward()
        return ro.steps, ro.host_s

    def teacher_part():
        (te.body(call, bufs, extras) * ml_weight).backward()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):                         # warm-up: tunes the GEMM shapes of the longer sampled rollouts, then the accumulate path
            sample_part(i)
            with hipops.Branch.like_capture():
                teacher_part()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    sample_part(0)                                 # the capture below must see every slice already written in this step
    g = torch.cuda.CUDAGraph()
    with _goat_graph(g):
        teacher_part()
    torch.cuda.synchronize()
    n, t_s, steps, host = 4, [], [], []
    t_all = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        st, hs = sample_part(i)
        torch.cuda.synchronize()
        t_s.append(time.perf_counter() - t0)
        steps.append(st)
        host.append(hs)
        bufs.load(te.plan(batches[i % len(batches)]))
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t_all) / n
    two = None
    if not os.environ.get('GOAT_BENCH_NO_TWO_PASS'):
        try:
            two = two_pass_iteration(call, te, bufs, g, batches, extras, arena, sim, store, ro, max_action_len, ml_weight)
        except Exception as e:      # noqa: BLE001
            two = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    forms = {'single_pass_eager': round(dt * 1e3, 2)}
    if isinstance(two, dict) and 'ms_per_iteration' in two:
        forms['two_pass'] = two['ms_per_iteration']
        if isinstance(two.get('single_pass_captured'), dict) and 'ms_per_iteration' in two['single_pass_captured']:
            forms['single_pass_captured'] = two['single_pass_captured']['ms_per_iteration']
        if isinstance(two.get('pass1_captured'), dict) and 'ms_per_iteration' in two['pass1_captured']:
            forms['two_pass_pass1_captured'] = two['pass1_captured']['ms_per_iteration']
            ov = two['pass1_captured'].get('teacher_overlapped')
            if isinstance(ov, dict) and 'ms_per_iteration' in ov:
                forms['two_pass_teacher_overlapped'] = ov['ms_per_iteration']
                pr = ov.get('pass1_on_a_high_priority_stream')
                if isinstance(pr, dict) and 'ms_per_iteration' in pr:
                    forms['two_pass_teacher_overlapped_pass1_high_priority'] = pr['ms_per_iteration']
    best = min(forms, key=forms.get)
    return {'best_form': best, 'best_ms_per_iteration': forms[best], 'forms_ms': forms,
            'ms_per_iteration': round(dt * 1e3, 2), 'sample_rollout_ms': round(sum(t_s) / n * 1e3, 2), 'sample_steps': round(sum(steps) / n, 1),
            'sample_host_builder_ms': round(sum(host) / n * 1e3, 2), 'teacher_part_ms': round((dt - sum(t_s) / n) * 1e3, 2),
            'ml_weight': ml_weight, 'max_action_len': max_action_len, 'two_pass': two,
            'what': 'teacher rollout (captured graph incl. its host plan, loss x %.1f, accumulate form) + sampled rollout (eager, one read-back per step) '
                    '+ their backward passes into one gradient arena: the reference iteration of train_alg=dagger' % ml_weight}


def two_pass_iteration(call, te, bufs, g_teacher, batches, extras, arena, sim, store, ro, max_action_len, ml_weight=0.2):
    """The same iteration with the sampled half in TWO passes: (1) the sampled rollout under no_grad, no loss — it only fixes
    the trajectory (rollout.NavRollout.actions; eager, one read-back per step); (2) TeacherEpisode.plan(actions=) re-walks it on the host
    with the DAgger labels and the episode graph captured at T = max_action_len replays forward + backward (dropout masks drawn anew);
    then the teacher graph in accumulate form as before.  The T = 15 graph runs the tuned GEMM configurations where the table has the
    shape and the heuristic ones elsewhere (the 15-step panorama batch): tuning them on first sight would add minutes to this leg."""
    from vln_goat_amd import hipops, rollout
    te_s = rollout.TeacherEpisode(sim, store, n_steps=max_action_len, text_len=te.L, pano_width=38, gmap_width=lambda t: 64)

    def pass1(i):
        hipops.RngState.dev.add_(0x9E3779B1)
        with torch.no_grad():
            ro.run(batches[i % len(batches)], feedback='sample', extras=extras, compute_loss=False)
        return ro.actions

    bufs_s = rollout.EpisodeBuffers(te_s.plan(batches[0], actions=pass1(0)))

    def sampled_body():
        arena.zero('nav')
        hipops.RngState.dev.add_(0x9E3779B1)
        te_s.body(call, bufs_s, extras).backward()
    tune = bool(os.environ.get('GOAT_BENCH_TUNE_TWO_PASS'))      # (one-off: measure the T = 15 shapes, then GOAT_SAVE_TUNED + scripts/merge_tuned.py)
    auto, hipops.AUTOTUNE = hipops.AUTOTUNE, hipops.AUTOTUNE and tune
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), hipops.Branch.like_capture():
            for _ in range(2):
                sampled_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_s = torch.cuda.CUDAGraph()
        with _goat_graph(g_s):
            sampled_body()
        torch.cuda.synchronize()
    finally:
        hipops.AUTOTUNE = auto
    n, t1, t2, t3, steps = 4, [], [], [], []
    for it in range(n + 1):                        # (first iteration untimed)
        batch = batches[it % len(batches)]
        t0 = time.perf_counter()
        acts = pass1(it)
        ta = time.perf_counter()
        plan = te_s.plan(batch, actions=acts)
        tb = time.perf_counter()
        bufs_s.load(plan)
        g_s.replay()
        bufs.load(te.plan(batch))
        g_teacher.replay()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        if it:
            t1.append(ta - t0), t2.append(tb - ta), t3.append(tc - tb), steps.append(ro.steps)
    ms = lambda v: round(sum(v) / len(v) * 1e3, 2)
    graphs = None
    try:
        # pass 1 as captured forward graphs (rollout.SampledEpisode): the step tables go into the SAME episode buffers as the walk proceeds,
        # the plan is finished by the time the last step has been sampled
        auto, hipops.AUTOTUNE = hipops.AUTOTUNE, hipops.AUTOTUNE and tune
        try:
            se = rollout.SampledEpisode(te_s, call, bufs_s, extras)
        finally:
            hipops.AUTOTUNE = auto
        import numpy as np
        rng = np.random.RandomState(17)
        u1, u2, usteps, uhost = [], [], [], []
        for it in range(n + 1):
            batch = batches[it % len(batches)]
            t0 = time.perf_counter()
            plan, _ = se.run(batch, rng)
            ta = time.perf_counter()
            bufs_s.load(plan)
            g_s.replay()
            bufs.load(te.plan(batch))
            g_teacher.replay()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            if it:
                u1.append(ta - t0), u2.append(tb - ta), usteps.append(se.steps), uhost.append(se.host_s)
        graphs = {'ms_per_iteration': round(ms(u1) + ms(u2), 2), 'pass1_step_graphs_ms': ms(u1), 'pass1_host_builder_ms': ms(uhost),
                  'sampled_graph_plus_teacher_part_ms': ms(u2), 'sample_steps': round(sum(usteps) / len(usteps), 1),
                  'what': 'pass 1 as %d captured forward graphs (instruction + one per step) fed step by step by EpisodePlanner / '
                          'EpisodeBuffers.load_part, one B x G read-back per step, sampling on the host; the finished plan feeds pass 2' % (max_action_len + 1)}
    except Exception as e:      # noqa: BLE001
        graphs = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    single = None
    if not os.environ.get('GOAT_BENCH_NO_SINGLE_PASS'):
        try:
            # ONE pass at graph speed (rollout.SinglePassSampledEpisode): step graphs captured with their autograd state + one captured backward graph;
            # the arena is cleared by the instruction graph, the teacher graph (accumulate form) follows as in the other forms
            import numpy as np
            auto, hipops.AUTOTUNE = hipops.AUTOTUNE, hipops.AUTOTUNE and tune
            try:
                sp = rollout.SinglePassSampledEpisode(te_s, call, bufs_s, extras, prologue=lambda: arena.zero('nav'))
            finally:
                hipops.AUTOTUNE = auto
            rng = np.random.RandomState(17)
            v1, v2, vsteps, vhost = [], [], [], []
            for it in range(n + 1):
                batch = batches[it % len(batches)]
                t0 = time.perf_counter()
                sp.run(batch, rng)
                ta = time.perf_counter()
                bufs.load(te.plan(batch))
                g_teacher.replay()
                torch.cuda.synchronize()
                tb = time.perf_counter()
                if it:
                    v1.append(ta - t0), v2.append(tb - ta), vsteps.append(sp.steps), vhost.append(sp.host_s)
            single = {'ms_per_iteration': round(ms(v1) + ms(v2), 2), 'sampled_walk_and_backward_launch_ms': ms(v1), 'host_builder_ms': ms(vhost),
                      'teacher_part_incl_backward_drain_ms': ms(v2), 'sample_steps': round(sum(vsteps) / len(vsteps), 1),
                      'what': 'the sampled half in ONE pass as the reference runs it (the action sampled from the forward that carries the loss): '
                              '%d forward graphs captured with their autograd state + one captured backward graph over all steps; then the teacher graph' % (max_action_len + 1)}
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            single = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    if isinstance(graphs, dict) and 'ms_per_iteration' in graphs and not os.environ.get('GOAT_BENCH_NO_OVERLAP'):
        try:
            graphs['teacher_overlapped'] = overlapped_iteration(call, te, te_s, bufs, bufs_s, batches, extras, arena, max_action_len, ml_weight, tune)
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            graphs['teacher_overlapped'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    return {'ms_per_iteration': round(ms(t1) + ms(t2) + ms(t3), 2), 'pass1_no_grad_rollout_ms': ms(t1), 'plan_along_actions_ms': ms(t2),
            'sampled_graph_plus_teacher_part_ms': ms(t3), 'sample_steps': round(sum(steps) / len(steps), 1), 'episode_bucket_T': max_action_len,
            'pass1_captured': graphs, 'single_pass_captured': single,
            'what': 'sampled rollout under no_grad (eager, fixes the trajectory) + host plan along the recorded actions + replay of the episode '
                    'graph captured at T = %d (forward + backward of the sampled half) + the teacher part as above' % max_action_len}


def overlapped_iteration(call, te, te_s, bufs, bufs_s, batches, extras, arena, max_action_len, ml_weight, tune):
    """The two-pass iteration with the TEACHER half moved into the shadow of pass 1.  Pass 1 is host-bound — per step the
    host builds the step's tables (1.4 ms), copies them, replays a forward graph, reads B x G probabilities back and samples — and leaves
    the GPU idle ~40 % of its 48 ms; the teacher rollout depends on nothing the policy samples.  So its captured graph (zero-first form:
    it clears the gradient arena and writes the teacher gradients x ml_weight) is launched on a side stream BEFORE pass 1 and runs in
    pass 1's gaps; the sampled half then replays in ACCUMULATE form behind it.  Same sum d(L_sample + ml_weight L_teacher) in the arena.
    Pass 1's step graphs are captured without the dropout-counter bump (rollout.SampledEpisode(bump_masks=False)): the counter must not
    move between the teacher graph's forward and backward kernels."""
    import numpy as np
    from vln_goat_amd import hipops, rollout

    def teacher_zero():
        arena.zero('nav')
        hipops.RngState.dev.add_(0x9E3779B1)
        (te.body(call, bufs, extras) * ml_weight).backward()

    def sampled_acc():
        hipops.RngState.dev.add_(0x9E3779B1)
        te_s.body(call, bufs_s, extras).backward()
    auto, hipops.AUTOTUNE = hipops.AUTOTUNE, hipops.AUTOTUNE and tune
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), hipops.Branch.like_capture():
            for _ in range(2):
                teacher_zero()
                sampled_acc()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        teacher_zero()                                 # (every slice written in this step: the capture below records the accumulate form)
        g_sa = torch.cuda.CUDAGraph()
        with _goat_graph(g_sa):
            sampled_acc()
        g_t0 = torch.cuda.CUDAGraph()
        with _goat_graph(g_t0):
            teacher_zero()
        se = rollout.SampledEpisode(te_s, call, bufs_s, extras, bump_masks=False)
        torch.cuda.synchronize()
    finally:
        hipops.AUTOTUNE = auto
    rng = np.random.RandomState(17)
    tstream = torch.cuda.Stream()
    hp = torch.cuda.Stream(priority=-1)                # pass 1 on a high-priority stream (second measurement below)
    ms = lambda v: round(sum(v) / len(v) * 1e3, 2)

    def measure(prio):
        n, tt, t_plan, t_p1, usteps = 4, [], [], [], []
        for it in range(n + 1):
            batch = batches[it % len(batches)]
            t0 = time.perf_counter()
            bufs.load(te.plan(batch))                      # the teacher's host plan + H2D
            ta = time.perf_counter()
            tstream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(tstream):
                g_t0.replay()                              # teacher forward + backward: runs beside pass 1
            if prio:                                       # the step graphs of pass 1 go ahead of the teacher graph's kernels wherever both are ready
                hp.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(hp):
                    plan, _ = se.run(batch, rng)
                torch.cuda.current_stream().wait_stream(hp)
            else:
                plan, _ = se.run(batch, rng)               # pass 1 (host-paced)
            tb = time.perf_counter()
            bufs_s.load(plan)
            torch.cuda.current_stream().wait_stream(tstream)
            g_sa.replay()                                  # sampled half, accumulating behind the teacher's gradients
            torch.cuda.synchronize()
            tc = time.perf_counter()
            if it:
                tt.append(tc - t0), t_plan.append(ta - t0), t_p1.append(tb - ta), usteps.append(se.steps)
        return tt, t_plan, t_p1, usteps
    pr = None
    if not os.environ.get('GOAT_BENCH_NO_PRIO'):
        try:
            ptt, _, pp1, _ = measure(True)
            pr = {'ms_per_iteration': ms(ptt), 'pass1_beside_teacher_graph_ms': ms(pp1)}
        except Exception as e:      # noqa: BLE001
            pr = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    tt, t_plan, t_p1, usteps = measure(False)
    return {'ms_per_iteration': ms(tt), 'teacher_plan_ms': ms(t_plan), 'pass1_beside_teacher_graph_ms': ms(t_p1),
            'pass1_on_a_high_priority_stream': pr,
            'sampled_graph_ms': round(ms(tt) - ms(t_plan) - ms(t_p1), 2), 'sample_steps': round(sum(usteps) / len(usteps), 1),
            'what': 'teacher plan + H2D, teacher graph (zero-first form, loss x %.1f) launched on a side stream, pass 1 as captured step graphs '
                    'beside it, then the sampled episode graph (T = %d) in accumulate form' % (ml_weight, max_action_len)}


def reverie_navigator_leg(args, ep, B, T):
    """The graph-only navigator on REVERIE observations (M/reverie/agent_obj_goat.py; BASELINE.json configs[4]'s dataset in fine-tuning): the
    fine-tuning model with the object-grounding head, 36 views + up to 20 objects per panorama (object features resident in HBM next to
    the view features), teacher forcing with the shortest-path expert, navigation + grounding loss, backward — NEW episodes every
    iteration: host plan in the worker process (object tables included), one pinned H2D, replay of the captured episode graph."""
    import numpy as np
    import time
    import vln_goat_amd
    from types import SimpleNamespace
    from vln_goat_amd import features, hipops, nav_model, rollout
    a = SimpleNamespace(num_l_layers=6, num_x_layers=3, num_pano_layers=2, dropout=0.1, feat_dropout=0.5, vocab_size=50265,
                        do_back_img=True, do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True,
                        do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door', mode='train', dataset='reverie',
                        obj_feat_size=768)
    torch.manual_seed(0)
    model = nav_model.GlocalTextPathNavCMT(nav_model.nav_config_from_args(a)).cuda().train()
    dt_ = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    rs = np.random.RandomState(57)
    scans = [rollout.ScanGraph.synthetic('rscan%d' % k, n=60, seed=70 + k, degree=3) for k in range(4)]
    keys = ['%s_%s' % (sc.name, vp) for sc in scans for vp in sc.vpids]
    store = features.FeatureStore.synthetic(keys, D=768, seed=5, dtype=dt_).to('cuda')
    O = 20
    objects = rollout.ObjectStore.synthetic(scans, D=768, max_objects=O, seed=9, dtype=dt_, p_empty=0.2).to('cuda')
    sim = rollout.GraphSim(store, objects=objects)
    L = ep['txt_ids'].shape[1]

    def batch_of_episodes(k):
        eps = []
        for b in range(B):
            sc = scans[(k + b) % len(scans)]
            dist, _ = sc.shortest()
            path = []
            for _ in range(64):
                s0 = int(rs.randint(len(sc.vpids)))
                far = int(np.argsort(dist[s0])[-1 - int(rs.randint(6))])
                cand = sc.shortest_path(sc.vpids[s0], sc.vpids[far])
                path = cand if len(cand) > len(path) else path
                if len(path) >= T:
                    break
            path = path[:T]                  # (the goal viewpoint is observed at step T - 1 at the latest: its grounding target is inside the episode)
            key = '%s_%s' % (sc.name, path[-1])
            ids = objects.attrs[key]['obj_ids'][:objects.count[key]]
            n_tok = int(rs.randint(L // 2, L - 1))
            eps.append({'instr_id': 'r%d_b%d' % (k, b), 'scan': sc, 'path': path, 'heading': float(rs.uniform(0, 2 * np.pi)),
                        'instr_encoding': [0] + rs.randint(3, 50000, n_tok - 2).tolist() + [2],
                        'obj_id': ids[int(rs.randint(len(ids)))] if len(ids) else None, 'end_vps': [path[-1]]})
        return eps
    batches = [batch_of_episodes(k) for k in range(6)]
    te = rollout.TeacherEpisode(sim, store, n_steps=T, text_len=L, pano_width=38, gmap_width=lambda t: 64, obj_width=O)
    extras = {'language': {k: ep[k] for k in ('instr_z_direction_features', 'instr_z_direction_pzs', 'instr_z_landmark_features',
                                              'instr_z_landmark_pzs', 'front_txt_feats')},
              'panorama': {'z_img_features': ep['z_img_features'], 'z_img_pzs': ep['z_img_pzs']},
              'navigation': {'front_txt_feats': ep['front_txt_feats'], 'front_vp_feats': ep['front_vp_feats'],
                             'front_gmap_feats': ep['front_gmap_feats']}}
    plan0 = te.plan(batches[0])
    n_og = sum(int((plan0['s%d_obj_target' % t] >= 0).sum()) for t in range(T))
    bufs = rollout.EpisodeBuffers(plan0)
    call = lambda mode, batch: model(mode, batch)
    params = list(model.parameters())

    arena = [None]

    def episode():
        if arena[0] is not None:
            arena[0].zero('nav')             # gradient arena: the weight gradients of the episode's Linears are deferred, merged per weight over the steps, grouped
        else:
            for p in params:
                p.grad = None
        hipops.RngState.dev.add_(0x9E3779B1)
        te.body(call, bufs, extras).backward()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            episode()
        if not args.no_arena:
            from vln_goat_amd import dp
            wrapper = dp.GoatDataParallel(model)
            wrapper.record_usage('nav')
            for p in params:
                p.grad = None
            arena[0] = wrapper.build_arena()
        with hipops.Branch.like_capture():
            for _ in range(2):
                episode()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _goat_graph(g):
        episode()
    pw = rollout.PlanWorker(te, store.keys, scans)
    state = {'submitted': 0, 'n_traj': [], 'wait': [], 'worker_s': []}

    def run(i):
        while pw.pending < 2:
            pw.submit(batches[state['submitted'] % len(batches)])
            state['submitted'] += 1
        t0 = time.perf_counter()
        plan = pw.result()
        state['wait'].append(time.perf_counter() - t0)
        pw.submit(batches[state['submitted'] % len(batches)])
        state['submitted'] += 1
        state['n_traj'].append(plan['_n_traj'])
        state['worker_s'].append(plan.get('_plan_s', 0.0))
        bufs.load(plan)
        g.replay()
    n = 12
    try:
        dt = timed(run, n, 3, 1)
    finally:
        while pw.pending:
            pw.result()
        pw.close()
    n_traj = sum(state['n_traj'][-n:])
    return {'ms_per_episode': round(dt / n * 1e3, 3), 'value': round(n_traj / dt, 1), 'unit': 'trajectory-steps/s', 'episodes': n,
            'host_plan_ms': round(sum(state['worker_s'][-n:]) / n * 1e3, 2), 'host_plan_wait_ms': round(sum(state['wait'][-n:]) / n * 1e3, 2),
            'grounding_targets_in_first_batch': n_og,
            'what': 'graph-only navigator on REVERIE observations: 4 synthetic scans (60 viewpoints, up to %d objects each), %d new episodes per '
                    'iteration, T = %d, 38 views + %d objects per panorama, map width 64, text bucket %d, teacher forcing (shortest-path expert), '
                    'navigation + object-grounding loss, fwd + bwd; host plan (worker process) + one pinned H2D + replay of the captured '
                    'episode graph' % (O, B, T, O, L)}


def navigator_leg(args, model, ep, arena, B, T, frozen_s):
    """The same rollout driven by the graph-only navigator: NEW episodes every iteration — start viewpoints, paths,
    instructions, panoramas, growing maps — on synthetic scans.  Per iteration, inside the timed region: the host walks the B
    ground-truth paths and builds every table of the T steps (rollout.TeacherEpisode.plan: candidates, maps with the reference's
    Floyd update, position features, logit-fusion matrices, targets, node-embedding gather indices), one pinned H2D moves them
    into the fixed-address episode buffers, and the captured episode graph (language, text K|V, T x (feature gather from the
    bf16 feature table resident in HBM, panorama, map gather, navigation), loss, backward) is replayed.  The plan of episode
    i + 1 is built while the GPU runs episode i."""
    import numpy as np
    from vln_goat_amd import features, hipops, rollout, synth
    rs = np.random.RandomState(31)
    scans = [rollout.ScanGraph.synthetic('scan%d' % k, n=60, seed=40 + k, degree=3) for k in range(4)]
    keys = ['%s_%s' % (sc.name, vp) for sc in scans for vp in sc.vpids]
    store = features.FeatureStore.synthetic(keys, D=768, seed=3, dtype=torch.bfloat16 if args.dtype == 'bf16' else torch.float32).to('cuda')
    sim = rollout.GraphSim(store)
    L = ep['txt_ids'].shape[1]

    def batch_of_episodes(k):
        eps = []
        for b in range(B):
            sc = scans[(k + b) % len(scans)]
            dist, _ = sc.shortest()
            path = []
            for _ in range(64):           # (a ground-truth path of >= T viewpoints where the scan has one: a 60-viewpoint scan's longest shortest
                s0 = int(rs.randint(len(sc.vpids)))          #  path is ~8 viewpoints, so at GOAT_NAV_T=15 the samples end early, as R2R's do under max_action_len 15)
                far = int(np.argsort(dist[s0])[-1 - int(rs.randint(6))])
                cand = sc.shortest_path(sc.vpids[s0], sc.vpids[far])
                path = cand if len(cand) > len(path) else path
                if len(path) >= T:
                    break
            n_tok = int(rs.randint(L // 2, L - 1))
            eps.append({'instr_id': 'k%d_b%d' % (k, b), 'scan': sc, 'path': path[:T + 2], 'heading': float(rs.uniform(0, 2 * np.pi)),
                        'instr_encoding': [0] + rs.randint(3, 50000, n_tok - 2).tolist() + [2]})
        return eps
    batches = [batch_of_episodes(k) for k in range(6)]
    te = rollout.TeacherEpisode(sim, store, n_steps=T, text_len=L, pano_width=38, gmap_width=lambda t: 64)
    extras = {'language': {k: ep[k] for k in ('instr_z_direction_features', 'instr_z_direction_pzs', 'instr_z_landmark_features',
                                              'instr_z_landmark_pzs', 'front_txt_feats')},
              'panorama': {'z_img_features': ep['z_img_features'], 'z_img_pzs': ep['z_img_pzs']},
              'navigation': {'front_txt_feats': ep['front_txt_feats'], 'front_vp_feats': ep['front_vp_feats'],
                             'front_gmap_feats': ep['front_gmap_feats']}}
    import time
    plans = [te.plan(batches[0])]              # (cold: shortest-path tables of the scans are built on first use)
    for k in (1, 2, 3):
        te.plan(batches[k])
    bufs = rollout.EpisodeBuffers(plans[0])
    call = lambda mode, batch: model(mode, batch)
    params = list(model.parameters())

    def episode():
        if arena is not None:
            arena.zero('nav')
        else:
            for p in params:
                p.grad = None
        hipops.RngState.dev.add_(0x9E3779B1)
        te.body(call, bufs, extras).backward()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), hipops.Branch.like_capture():
        for _ in range(2):
            episode()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _goat_graph(g):
        episode()
    # the host plan of episode i + 1 is built by a WORKER PROCESS (rollout.PlanWorker) while this process copies plan i into the pinned
    # buffer and launches the episode graph (a ~3 000-node hipGraphLaunch holds the calling thread for several ms): with the navigation
    # step's parallel branches the replay takes 19 ms, and plan (13 ms) + copy + launch behind one another no longer fit under it (a
    # worker THREAD was measured too: 21.4-24.3 ms per episode — the table builders hold the GIL)
    _progress('navigator: teacher episode captured')
    pw = rollout.PlanWorker(te, store.keys, scans)
    state = {'submitted': 0, 'plan_s': [], 'n_traj': []}

    def run(i):
        while pw.pending < 2:               # two plans in flight: the one about to be consumed was submitted two iterations ago
            pw.submit(batches[state['submitted'] % len(batches)])
            state['submitted'] += 1
        t0 = time.perf_counter()
        plan = pw.result()
        state['plan_s'].append(time.perf_counter() - t0)                      # what this process WAITS for the plan (0 when it is hidden)
        pw.submit(batches[state['submitted'] % len(batches)])               # host work of the next episode, in the worker
        state['submitted'] += 1
        state['n_traj'].append(plan['_n_traj'])
        state.setdefault('worker_s', []).append(plan.get('_plan_s', 0.0))
        bufs.load(plan)                      # pinned H2D, enqueued behind the previous replay
        g.replay()
    n = 12
    dt = timed(run, n, 3, 1)
    while pw.pending:
        pw.result()
    pw.close()
    in_loop = state['plan_s'][-n:]
    wait_ms = sum(in_loop) / len(in_loop) * 1e3          # mean time the training process waited for a plan inside the timed loop
    plan_ms = sum(state['worker_s'][-n:]) / n * 1e3      # what one plan costs: measured around TeacherEpisode.plan inside the worker
    n_traj = sum(state['n_traj'][-n:])
    _progress('navigator: teacher loop timed; DAgger iteration')
    dagger = None
    if arena is not None and not os.environ.get('GOAT_BENCH_NO_DAGGER'):
        try:
            dagger = dagger_iteration(model, te, bufs, batches, extras, arena, sim, store, B, T)
        except Exception as e:      # noqa: BLE001
            dagger = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    return {'dagger_iteration': dagger, 'ms_per_episode': round(dt / n * 1e3, 3), 'value': round(n_traj / dt, 1), 'unit': 'trajectory-steps/s', 'episodes': n,
            'vs_frozen_episode': round((dt / n) / frozen_s, 3), 'host_plan_ms': round(plan_ms, 2), 'host_plan_wait_ms': round(wait_ms, 2), 'host_plan_wait_ms_max': round(max(in_loop) * 1e3, 2), 'h2d_bytes_per_episode': bufs.nbytes,
            'what': 'graph-only navigator on 4 synthetic scans (60 viewpoints each), %d new episodes per iteration, teacher forcing, '
                    'pano width 38, map width 64, text bucket %d; host plan (worker process) + one pinned H2D + replay of the captured episode graph' % (B, L)}


def main():
    args = parse()
    rc = spawn_ranks(args)
    if rc is not None:
        sys.exit(rc)
    if os.environ.get('GOAT_BENCH_LAUNCH_ONLY'):
        return launch_check(args)
    if args.leg:
        torch.cuda.set_device(0)
        print(json.dumps({'config5': config5_leg, 'config4': config4_leg, 'large_batch': large_batch_leg}[args.leg](args)))
        if os.environ.get('GOAT_SAVE_TUNED'):
            from vln_goat_amd import hipops
            hipops.save_tuned(os.environ['GOAT_SAVE_TUNED'] + '.' + args.leg)
        return
    world, rank, local = setup_dist(args)
    if args.workload == 'config4':
        # BASELINE.json configs[3]: the fine-tuning iteration, data-parallel (vln_bert + critic wrapped as M/r2r/agent_base.py:100-102)
        r = config4_leg(args, rank, world)
        if rank == 0:
            print(json.dumps({
                'metric': 'trajectory-steps/sec fwd+bwd (GOAT fine-tune rollout, 36 views x 768, 200 tok)', 'value': r['value'], 'unit': r['unit'],
                'n_gpus': world, 'steps': r['episodes'], 'warmup': args.warmup, 'ms_per_step': r['ms_per_episode'], 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
                'config': {'workload': r['workload'], 'name': 'BASELINE.json configs[3]', 'global_batch': r['per_rank_batch'] * world,
                           'parallelism': 'dp%d' % world, 'launch': r['launch'], 'wire': args.wire},
                'roofline': r['roofline'], 'dp': r['dp']}))
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    wl = WORKLOADS[args.workload]
    per_rank = args.batch or wl['per_rank']
    m = measure_pretrain(args, world, rank, args.workload, args.steps, args.warmup)
    dt, n_traj, wrapper = m['dt'], m['n_traj'], m['wrapper']

    # each task's step alone (same graphs, 12 replays each, all ranks take part: the steps hold collectives at N > 1)
    per_task = {}
    for t in ([] if os.environ.get('GOAT_BENCH_NO_PER_TASK') else m['tasks']):      # (off in the counter passes: their window is the tail of the run)
        n_t = 12
        dt_t = timed(lambda i, t=t: m['steps'][t](), n_t, 2, world)
        per_task[t] = round(dt_t / n_t * 1e3, 3)
    dp_diag = None
    if world > 1 and not os.environ.get('GOAT_BENCH_NO_DP_DIAG'):
        try:
            dp_diag = dp_diagnostics(args, m, world, rank, dt)
        except Exception as e:      # noqa: BLE001  (never lose the scaling point to the diagnostics)
            dp_diag = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0:
        value = n_traj * world * args.steps / dt
        algo = sum(ALGO_GFLOP_PER_TRAJ_STEP.values()) / 3.0
        out = {
            'metric': 'trajectory-steps/sec fwd+bwd (GOAT pretrain, 36 views x 768, %d tok)' % wl['batch']['L'],
            'value': round(value, 1), 'unit': 'trajectory-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': wl['text'] % {'layers': args.layers, 'batch': per_rank}
                                   + ', fwd+bwd%s, random-init' % (' + grad all-reduce' if world > 1 else ''),
                       'name': 'BASELINE.json configs[%d]' % {'config2': 1, 'config5': 4}[args.workload],
                       'global_batch': per_rank * world, 'parallelism': 'dp%d' % world, 'wire': args.wire,
                       'launch': 'eager' if args.no_graph else ('hipGraph replay' if world == 1 else
                                                                      'hipGraph replay, backward cut into %d phases whose gradient all-reduces overlap the later phases (cfp: one more cut around the eager all-gather + loss)' % wrapper.n_phases
                                                                      if wrapper.launch_mode == 'phased' else
                                                                      'ONE hipGraph per step: %d backward phases, each gradient exchange a parallel branch of the graph' % wrapper.n_phases
                                                                      if wrapper.launch_mode == 'in-graph' else 'hipGraph replay of forward + backward, then one gradient all-reduce (fallback path)')},
            'samples_per_s': round(value / 5.0, 1),
        }
        if dp_diag is not None:
            out['dp'] = dp_diag
        headline = args.workload == 'config2' and TASKS == ('mlm', 'sap', 'cfp')
        if headline:
            out['step_mfma_frac'] = round(value / world * algo * 1e9 / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12), 4)
            out['ms_per_task_step'] = per_task
        if world == 1 and headline and not args.no_roofline:
            out['roofline'] = gemm_roofline(args, m['model'], m['gb'], wrapper.arena)
        if world == 1 and headline and not args.no_extra_configs and not args.no_graph and not args.no_arena:
            out['fresh_batch'] = leg('fresh_batch', lambda: fresh_batch_leg(args, m))
            if 'ms_per_step' in out['fresh_batch']:
                out['fresh_batch_ms_per_step'] = out['fresh_batch']['ms_per_step']
            tl = (out['fresh_batch'].get('ragged_bucketed') or {}).get('train_loop')
            if tl:                                   # the whole training loop (new ragged batch + fwd/bwd + clip + AdamW), next to the headline
                out['train_loop'] = tl
            out['with_optimizer'] = leg('with_optimizer', lambda: optimizer_leg(args, m))
        from vln_goat_amd import hipops as _h
        out['gemm_shapes_autotuned_in_this_run'] = _h.TUNE_EVENTS[0]      # 0: every shape came from vln-goat_amd/tuned_gfx950.json
        if os.environ.get('GOAT_SAVE_TUNED'):      # persist the autotuned GEMM table (copied to vln-goat_amd/tuned_gfx950.json)
            from vln_goat_amd import hipops
            hipops.save_tuned(os.environ['GOAT_SAVE_TUNED'])
        cfg = m['cfg']
        if world == 1 and headline and not args.no_extra_configs and not os.environ.get('GOAT_BENCH_ONLY_FRESH'):
            m.clear()
            out['config5_reverie'] = leg_process(args, 'config5')
            out['config4_nav'] = leg_process(args, 'config4')
            if not os.environ.get('GOAT_BENCH_NO_LARGE_BATCH'):
                out['large_batch'] = leg_process(args, 'large_batch')
        if world == 1 and headline and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, cfg)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Reproducer of the hipGraph lifetime crash hipops._retain_cuda_graphs works around (ROCm 7.2): three graph captures in one
process, the first two destroyed -> hipGraphLaunch of the third segfaults in hip::Graph::UpdateStreams.
    GOAT_NO_GRAPH_RETAIN=1 python scripts/dbg_graph_lifetime.py noopt,noeager,noreplay,task_mlm     -> Segmentation fault
    python scripts/dbg_graph_lifetime.py noopt,noeager,noreplay,task_mlm                            -> A ok, B ok, C ok"""
import os, sys, faulthandler, gc
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
faulthandler.enable()
import torch
import vln_goat_amd
from vln_goat_amd import config as gcfg, hipops, optim, pretrain_model, synth, train_step, dp
import test_train_step_gpu as T
P = lambda *a: print(*a, flush=True)
flags = set(sys.argv[1].split(',')) if len(sys.argv) > 1 else set()
if 'keep' in flags:
    _KEEP = []
    _orig = torch.cuda.CUDAGraph
    class _K(_orig):
        def __new__(cls, *a, **k):
            o = super().__new__(cls, *a, **k)
            _KEEP.append(o)
            return o
    torch.cuda.CUDAGraph = _K

def A():
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=1, num_pano_layers=1, vocab_size=1000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
    gb = synth.batch_to(synth.make_pretrain_batch(B=4, T=[2, 3, 1, 2], L=[30, 22, 16, 25], seed=5, vocab_size=1000, style='rich'), 'cuda')
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    if 'noarena' in flags:
        params = list(model.parameters())
        def body(task):
            for p in params:
                p.grad = None
            model(gb, task, compute_loss=True).mean().backward()
        flags.add('noopt'); flags.add('noeager')
    else:
        wrapper, arena = T._arena_for(model, gb)
        opt = optim.FusedAdamW(model.named_parameters(), arena, lr=2e-3, betas=(0.9, 0.98), weight_decay=0.01)
        def body(task):
            arena.zero(task)
            model(gb, task, compute_loss=True).mean().backward()
    if 'nograph' in flags:
        for t in ('mlm', 'sap', 'cfp'):
            body(t)
            if 'noopt' not in flags:
                opt.step(t, max_norm=5.0)
        torch.cuda.synchronize()
        vln_goat_amd.set_compute_dtype(torch.float32)
        return
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for t in ('mlm', 'sap', 'cfp'):
            body(t)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs = {}
    TT = [x[5:] for x in flags if x.startswith('task_')] or ['mlm', 'sap', 'cfp']
    for t in TT:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(t)
        graphs[t] = g
    for rnd in range(0 if 'noreplay' in flags else 2):
        for t in TT:
            graphs[t].replay()
            if 'noopt' not in flags:
                opt.step(t, max_norm=5.0)
    torch.cuda.synchronize()
    if 'noeager' not in flags:
        for t in TT:
            graphs[t].replay()
            by_id = {id(p): p for p in model.parameters()}
            if 'norefresh' not in flags:
                for p in model.parameters():
                    hipops.refresh_shadows(p, by_id)
            arena.zero(t)
            model(gb, t, compute_loss=True).mean().backward()
            torch.cuda.synchronize()
    vln_goat_amd.set_compute_dtype(torch.float32)

if 'skipA' not in flags:
    A(); gc.collect(); P('A ok')
if 'skipB' not in flags:
    T.test_static_batch_feeds_a_captured_step_with_new_batches(); gc.collect(); P('B ok')
T.test_shape_bucketed_static_batch_replays_ragged_batches(); P('C ok')

"""2-rank self-check of the data-parallel FINE-TUNING iteration (BASELINE configs[3]) with the real HIP navigation model; two ranks
share the one GPU of the test box, collectives through gloo (RCCL refuses two ranks per device):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29549 scripts/dp_nav_check.py
dp.wrap_finetune_models (vln_bert + critic, M/r2r/agent_base.py:100-102) + the gradient arena: each rank rolls out its half of a 4-sample
teacher-forced episode (language once, 3 x (panorama, navigation), BPTT through [MEM]), loss / local batch, one backward, gradient average —
must equal ONE process on the 4 samples (loss / 4): float32 wire to 1e-4 of every tensor's scale, bf16 wire to 2e-2."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from types import SimpleNamespace
import torch
import torch.distributed as dist
import vln_goat_amd
from vln_goat_amd import dp, nav_model, synth, hipops

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
a = SimpleNamespace(num_l_layers=2, num_x_layers=2, num_pano_layers=1, dropout=0.0, feat_dropout=0.0, vocab_size=1000,
                    do_back_img=True, do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True,
                    do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door', mode='train')
torch.manual_seed(0)
model = nav_model.GlocalTextPathNavCMT(nav_model.nav_config_from_args(a)).cuda().eval()      # (eval: the config's own dropout rates off)
critic = nav_model.Critic(a).cuda().eval()
if rank == 1:                                   # the wrapper must broadcast rank 0's weights
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01)
vln_goat_amd.set_compute_dtype(torch.float32)
hipops.manual_seed(99)
B = 4
full = synth.make_nav_episode(B=B, L=40, n_steps=3, seed=7, vocab_size=1000, extra_nodes=3)


def shard(x, lo, hi):
    if torch.is_tensor(x):
        return x[lo:hi] if x.dim() > 0 and x.shape[0] == B else x
    if isinstance(x, list) and len(x) == B:
        return x[lo:hi]
    return x


def cut(ep, lo, hi):
    return {k: ([{kk: shard(vv, lo, hi) for kk, vv in st.items()} for st in v] if k == 'steps' else shard(v, lo, hi)) for k, v in ep.items()}


per = B // world
mine = cut(full, rank * per, (rank + 1) * per)
ok = True
worst_all = {}
for wire in (None, torch.bfloat16):
    for p in list(model.parameters()) + list(critic.parameters()):
        p.grad = None
    w, wc = dp.wrap_finetune_models(model, critic, wire_dtype=wire)

    def iteration(ep, n):
        loss, _ = synth.run_nav_episode(lambda m, b: w(m, b), ep, device='cuda')
        return loss / n
    iteration(mine, per).backward()
    w.record_usage('nav')
    for p in model.parameters():
        p.grad = None
    arena = w.build_arena()
    for _ in range(2):
        arena.zero('nav')
        loss = iteration(mine, per)
        loss.backward()
        dp.reduce_finetune_gradients((w, wc))
    torch.cuda.synchronize()
    got = {n: arena.views[id(p)].clone() for n, p in model.named_parameters() if id(p) in arena.views}
    arena.detach()
    # one process, all four samples, plain autograd gradients
    for p in model.parameters():
        p.grad = None
    ref_loss = iteration(full, B)
    ref_loss.backward()
    torch.cuda.synchronize()
    lsum = torch.tensor([float(loss)], dtype=torch.float64)
    dist.all_reduce(lsum)
    tol = 1e-4 if wire is None else 2e-2
    assert abs(float(lsum) / world - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss))), (float(lsum) / world, float(ref_loss))
    top = max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)
    worst, n_cmp = 0.0, 0
    for n, p in model.named_parameters():
        if p.grad is None:
            assert n not in got or float(got[n].abs().max()) == 0.0, n
            continue
        assert n in got, n
        scale = max(float(p.grad.abs().max()), 1e-3 * top)
        worst = max(worst, float((got[n] - p.grad).abs().max()) / scale)
        n_cmp += 1
    assert all(p.grad is None for p in critic.parameters())
    worst_all['f32' if wire is None else 'bf16'] = worst
    if rank == 0:
        print('wire %s: %d parameter gradients, worst deviation from the single-process iteration %.2e of the tensor scale (bound %.0e)'
              % ('f32' if wire is None else 'bf16', n_cmp, worst, tol))
    ok = ok and worst < tol and n_cmp > 50
ok = ok and worst_all['bf16'] > 0
flag = torch.tensor([int(ok)]); dist.all_reduce(flag)
if rank == 0:
    print('DP_NAV_CHECK_OK' if int(flag) == world else 'DP_NAV_CHECK_FAILED')
dist.barrier(); dist.destroy_process_group()

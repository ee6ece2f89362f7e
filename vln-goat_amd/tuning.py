"""GEMM tile configuration of goat_gemm_bf16 / goat_wgrad_grouped: the candidate tiles, the static heuristic, the table of measured
configurations (tuned_gfx950.json) and the autotuner that fills it on first sight of a shape.  (Round 6: moved out of hipops.py —
VERDICT r5 #7: round 5's largest gain was a silent plan miss in this plumbing; `STATS` below counts every launch that runs on a
configuration nobody measured, and tests/test_train_step_gpu.py fails when a captured step contains one.)

State that callers switch lives HERE (`tuning.AUTOTUNE = True`, `tuning.PROFILE = []`); `hipops.AUTOTUNE` / `hipops.PROFILE` keep working
as aliases (hipops forwards reads and writes of the two names to this module)."""
import json
import os

import torch

from . import _lib
from ._lib import EPI_NONE
from ._plumbing import _dt, _ptr, _stream

PROFILE = None   # bench.py sets this to a list to time every GEMM launch with HIP events (hipops.gemm / gemm_nt, WgradQueue._launch)

# launches on a configuration that was never measured: 'gemm_heuristic' = a goat_gemm_bf16 shape missing from the table (static heuristic),
# 'wgrad_default' = a weight-gradient group without a timed plan (WgradQueue.cfg).  Counted always; `STATS_LOG` keeps the keys of the misses.
STATS = {'gemm_heuristic': 0, 'gemm_tuned': 0, 'wgrad_default': 0, 'wgrad_tuned': 0}
STATS_LOG = []


def reset_stats():
    for k in STATS:
        STATS[k] = 0
    del STATS_LOG[:]


AUTOTUNE = False        # bench.py / trainers may switch this on: first sight of a shape times the candidate configs
_TUNED = {}             # (ta, tb, M, N, Kc, epi, f32out, split_req) -> (bm, nstage, split)
_FLUSH = [None]
TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned_gfx950.json')


def load_tuned(path=None):
    """Merge a saved table of autotuned GEMM configurations (measured on an MI355X by bench.py) into _TUNED."""
    path = path or os.environ.get('GOAT_TUNED_FILE') or TUNED_FILE
    if os.environ.get('GOAT_NO_TUNED') or not os.path.exists(path):      # GOAT_NO_TUNED=1: re-tune from scratch (bench.py GOAT_SAVE_TUNED=...)
        return 0
    with open(path) as f:
        tab = json.load(f)
    for k, v in tab.items():
        kk = json.loads(k)
        _TUNED.setdefault(tuple(bool(x) if i in (0, 1, 6, 8) else int(x) for i, x in enumerate(kk)), tuple(int(x) for x in v))
    return len(tab)


def save_tuned(path):
    tab = {json.dumps([int(x) for x in k]): list(v) for k, v in sorted(_TUNED.items())}
    with open(path, 'w') as f:
        json.dump(tab, f, indent=0, sort_keys=True)
    return len(tab)



EIGHT_WAVES = 0x100     # GOAT_GEMM_8WAVES (include/goat_hip.h): flag in the nstage argument of goat_gemm_bf16
PINGPONG = 0x200        # GOAT_GEMM_PP: the ping-pong main loop (csrc/gemm5_tile.hpp); tiles 256x256, 192x256, 128x256, 256x128, 128x128
BALANCED = 0x800        # (grouped weight gradients only) goat_wgrad_grouped_balanced: one workgroup per CU, equal shares of the group's K-tile iterations
PERSIST = 0x400         # GOAT_GEMM_PERSIST (with PINGPONG): one workgroup per CU walks the tiles, next tile's first K-tile requested before the epilogue
USE_PP = os.environ.get('GOAT_GEMM_NO_PP', '0') == '0'
USE_PERSIST = os.environ.get('GOAT_GEMM_NO_PERSIST', '0') == '0'
N_CU = 256              # MI355X; `n_cu()` reads the device (decides which shapes get the persistent candidates timed, and the tail split of a group)


def n_cu():
    """compute units of the current device (the kernels read the same attribute: pp_cu_count())"""
    global N_CU
    if not _N_CU_READ[0] and torch.cuda.is_available():
        N_CU = int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count) or N_CU
        _N_CU_READ[0] = True
    return N_CU


_N_CU_READ = [False]


def tile(bm, bn=128):
    """tile argument of goat_gemm_bf16 / goat_wgrad_grouped: rows | columns << 16 (128 columns: just the row count)."""
    return bm if bn == 128 else (bm | (bn << 16))


def tile_name(t):
    return '%dx%d' % (t & 0xFFFF, (t >> 16) or 128)


def stage_name(ns):
    return ('pp' if ns & PINGPONG else 's%d' % (ns & 0xFF)) + ('8w' if ns & EIGHT_WAVES else '') + ('P' if ns & PERSIST else '') + ('B' if ns & BALANCED else '')


def _tile_candidates(ta, tb, M, N):
    """(tile, ring stages) pairs the autotuner times for one GEMM shape.  The 8-wave 192/256-wide tiles (csrc/gemm3.hip) need a
    power-of-two width on a transposed operand's side; they only pay when the problem has enough rows / columns."""
    c = [(64, 2), (64, 3), (64, 4), (128, 2), (128, 3), (128, 4), (128, EIGHT_WAVES | 2), (128, EIGHT_WAVES | 3), (128, EIGHT_WAVES | 4)]
    if M >= 2048:
        c += [(256, 2), (256, 3)]
    if not ta and M >= 960:              # 96-row tiles: 3840 / 96 = 40 tile rows -> 240 tiles at N = 768 (one round on 256 CUs)
        c += [(96, 2), (96, 3), (96, 4)]
    if N >= 256 and M >= 512:
        c += [(tile(128, 256), 2), (tile(128, 256), 3)]
        if M >= 1024:
            c += [(tile(256, 256), 2)]
            if not ta:
                c += [(tile(192, 256), 2)]
    if N >= 384 and M >= 1024 and not ta and not tb:
        c += [(tile(256, 192), 2), (tile(192, 192), 2), (tile(192, 192), 3)]
    if USE_PP and M >= 512 and N >= 256:          # ping-pong main loop (eight waves, >= 128 x 128 tiles)
        c += [(128, PINGPONG | 2), (tile(128, 256), PINGPONG | 2), (256, PINGPONG | 2)]
        if M >= 1024:
            c += [(tile(256, 256), PINGPONG | 2)]
            if not ta:
                c += [(tile(192, 256), PINGPONG | 2)]
        if USE_PERSIST and not ta:                # the persistent form of the same tiles where a problem has more tiles than workgroup slots
            for t, ns in [x for x in c if x[1] & PINGPONG]:
                rows, cols = t & 0xFFFF, (t >> 16) or 128
                slots = n_cu() * (2 if rows * cols <= 128 * 128 else 1)
                if ((M + rows - 1) // rows) * ((N + cols - 1) // cols) > slots:
                    c.append((t, ns | PERSIST))
    return c


def _heuristic_cfg(ta, tb, M, N, Kc, split_k):
    """(bm, nstage) measured with scripts/gemm_bench.py (hot and cold operands)."""
    tiles128 = ((M + 127) // 128) * ((N + 127) // 128) * max(1, split_k)
    kper = Kc // max(1, split_k)
    if ta:
        return 64, 2
    if kper >= 2048 and tiles128 >= 150:
        return 128, 3
    if tiles128 >= 1024:
        return 128, 2
    return 64, 2


def _launch_gemm_bf16(a, b, out, ta, tb, M, N, Kc, bias, epi, aux, split_k, bm, nstage, colsum_out):
    args = (_stream(), int(ta), int(tb), _dt(out), _ptr(a), a.stride(0), _ptr(b), b.stride(0),
            _ptr(out), out.stride(0), M, N, Kc,
            _ptr(bias) if bias is not None else None, epi,
            _ptr(aux) if aux is not None else None,
            aux.stride(0) if aux is not None else 0, split_k, bm, nstage,
            _ptr(colsum_out) if colsum_out is not None else None)
    st = _lib.lib().goat_gemm_bf16(*args)
    _lib.check(st, 'goat_gemm_bf16(ta=%d,tb=%d,M=%d,N=%d,Kc=%d,bm=%d,ns=%d,split=%d)' % (ta, tb, M, N, Kc, bm, nstage, split_k))


def _time_cfg(fn, reps=4):
    """median HIP-event time of fn() with the L2 / Infinity Cache flushed before every repetition (inside a
    training step the operands of a GEMM are cold: they were just produced by another kernel)."""
    if _FLUSH[0] is None:
        _FLUSH[0] = torch.empty(320 << 20, dtype=torch.uint8, device='cuda')
    ts = []
    for _ in range(reps):
        _FLUSH[0].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


TUNE_EVENTS = [0]        # shapes timed by the autotuner in this process (0 when tuned_gfx950.json covers the run: bench.py reports it)


def _tune_gemm(key, a, b, out, ta, tb, M, N, Kc, bias, epi, aux, split_opts, colsum_out):
    TUNE_EVENTS[0] += 1
    kt = (Kc + 63) // 64
    best = None
    scratch = torch.empty_like(out) if out.dtype == torch.float32 else out
    cs = torch.zeros_like(colsum_out) if colsum_out is not None else None
    for split in split_opts:
        if split > kt:
            continue
        for bm, ns in _tile_candidates(ta, tb, M, N):
            if bm == 128 and (ns & 0xFF) == 4 and out.dtype == torch.bfloat16 and epi != EPI_NONE:
                continue
            try:
                t = _time_cfg(lambda: _launch_gemm_bf16(a, b, scratch, ta, tb, M, N, Kc, bias, epi, aux, split, bm, ns, cs))
            except RuntimeError:
                continue
            if split > 1:       # a split launch needs its float32 output cleared first: count that fill (ms)
                t += 1.5e-3 + out.numel() * 4 / 4.0e9
            if best is None or t < best[0]:
                best = (t, bm, ns, split)
    _TUNED[key] = best[1:]
    return best[1:]

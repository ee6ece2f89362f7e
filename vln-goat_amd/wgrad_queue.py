"""Deferred, grouped weight gradients (`WgradQueue`) and the deferred LayerNorm column reduction (`LnReduceQueue`) of the backward pass.
(Round 6: moved out of hipops.py; `hipops.WgradQueue` / `hipops.LnReduceQueue` are the same classes.)"""
import ctypes
import os

import torch

from . import _lib, tuning
from ._plumbing import _ptr, _stream
from .streams import Branch
from .tuning import BALANCED, EIGHT_WAVES, PINGPONG, USE_PP, _time_cfg, n_cu, stage_name, tile, tile_name


class WgradQueue:
    """Deferred weight gradients.  With a gradient arena attached the weight gradient of a Linear is not needed until the
    backward pass ends, so instead of launching each small dW = dY^T·X on its own (36-144 tiles: split along the
    contraction, atomics and a zero fill to occupy 256 CUs) the problems are queued and executed up to sixteen at a time by
    goat_wgrad_grouped: one unsplit launch that fills the chip.  Flushed when full, when a queued parameter is about
    to be written again (ordering), and by an autograd-engine callback at the end of the backward pass.
    The first write of a slice in a step overwrites it; a later write (shared weights, BPTT) is queued as an accumulation —
    never in the same group as an earlier write of that slice (hipops._sink flushes first)."""
    enabled = os.environ.get('GOAT_WGRAD_GROUP', '1') != '0'
    # (tile = rows | cols << 16, ring stages | 0x100 = eight waves on 128x128 | 0x200 = ping-pong): scripts/wgrad_group_bench.py.  The configuration of a
    # group the tuner has not timed — above all the groups of a CAPTURED step when its eager warm-up ran on one stream: a capture forks parallel
    # branches, each stream has its own queue, so the groups differ from the warm-up's and miss the tuned plans (found in round 5: 10 of the 12
    # groups of the headline cycle ran this default, then 256 x 128 on three ring slots; the 256 x 256 ping-pong tile wins nearly every group the
    # tuner times: step 5.30 -> 5.18 ms same box, profiles/round5_wgrad_default_cfg_ab.txt).  Branch.like_capture() makes a warm-up form the capture's groups.
    cfg = tuple(int(v) for v in os.environ['GOAT_WGRAD_GROUP_CFG'].split(',')) if 'GOAT_WGRAD_GROUP_CFG' in os.environ else (
        tuple(int(v) for v in os.environ['GOAT_WGRAD_DEFAULT_CFG'].split(',')) if 'GOAT_WGRAD_DEFAULT_CFG' in os.environ else (   # (A/B: default without switching the tuner off)
            (256 | 256 << 16, 0x200 | 2) if USE_PP else (256, 3)))
    MAX = int(os.environ.get('GOAT_WGRAD_GROUP_MAX', '48'))      # problems per launch = the kernel's GROUP_MAX (48 since the last session of round 5: 24 / 32 / 48 -> 5.03 / 5.02 / 4.99 ms per step; round 1: 8 / 12 / 16 -> 7.12 / 7.09 /
                                                                 # 7.06 ms per step; round 4, same box, three alternations: 16 -> 5.80 / 5.81 / 5.80, 24 -> 5.75 / 5.76 / 5.76)
    # (round 2 also had a mode that ran the grouped launches on a stream of their own, off the dgrad chain: 6.52 vs 6.28 ms per step — the
    #  kernels contend, they do not fill idle CUs: removed in round 3)
    # (diagnostics; measured and NOT adopted) every queued problem waits for the end of the backward pass (or for a re-write of its slice)
    # and the launches then run back to back, MAX problems each — instead of interleaving with the dgrad chains, whose short kernels on
    # the OTHER graph branch starve behind a chip-filling grouped launch (profiles/round4_step_ln_attention_by_shape.txt: 10 us kernels
    # stretched to 200 us).  Same step time (5.80 / 5.80 / 5.82 vs 5.80 / 5.81 / 5.80 ms): what the chains gain, the lost overlap costs.
    DEFER_ALL = os.environ.get('GOAT_WGRAD_DEFER_ALL', '0') == '1'
    queues = {}             # HIP stream handle -> (torch stream, [(dy, x, w_sink, b_sink, accumulate)]): tensors are kept alive until
    pending_ids = {}        # the launch, which happens on the stream the problems were produced on;  id(param) -> stream handle
    _callback_armed = False
    # Merging (last session of round 5).  A weight used several times in one backward pass — every weight of the navigation model in a
    # T-step episode (BPTT), shared cross-modal layers — used to force a flush at its second use (two problems writing one slice cannot share
    # a launch), so an episode's backward ran one grouped launch PER STEP, each reading and re-writing the float32 gradient of every weight:
    # 16 launches of ~180 us per 6-step episode, bound by that traffic, not by MFMA (contraction lengths of 12 ... 768 rows).  Now small
    # problems of the same slice stay queued together and become ONE problem at launch time: dW = [dY_1; dY_2; ...]^T [X_1; X_2; ...] (the
    # row blocks copied into one buffer each — cheap while rows < 2 n_out n_in / (n_out + n_in), the copy against a read-modify-write of dW).
    MERGE = os.environ.get('GOAT_WGRAD_MERGE', '1') != '0'
    distinct = {}           # HIP stream handle -> set of queued slices (data pointers): the flush threshold counts problems AFTER merging
    MAX_ITEMS, MAX_SLICES = 8192, 1024        # (bounds on queued tensors / distinct slices)

    @classmethod
    def mergeable(cls, rows, n_out, n_in):
        return cls.MERGE and cls.enabled and rows * (n_out + n_in) < 2 * n_out * n_in and n_out % 8 == 0 and n_in % 8 == 0      # (the concatenated operands are contiguous: row length = leading dimension, a multiple of 8)

    @classmethod
    def push(cls, dy, x, w_sink, b_sink, param_ids, accumulate):
        st = torch.cuda.current_stream()
        q = cls.queues.setdefault(st.cuda_stream, (st, []))[1]
        q.append((dy, x, w_sink, b_sink, int(bool(accumulate))))
        d = cls.distinct.setdefault(st.cuda_stream, [set(), set()])      # [every queued slice, slices with a problem too large to merge]
        d[0].add(w_sink.data_ptr())
        if not cls.mergeable(dy.shape[0], dy.shape[1], x.shape[1]):
            d[1].add(w_sink.data_ptr())
        for i in param_ids:
            cls.pending_ids[i] = st.cuda_stream
        cls.arm()
        # full = MAX slices that will not be merged with later problems; the small (mergeable) ones wait for the other steps' problems of
        # their weight — an episode's backward is then a handful of launches at its end instead of one per step
        if (len(d[1]) >= cls.MAX and not cls.DEFER_ALL) or len(d[0]) >= cls.MAX_SLICES or len(q) >= cls.MAX_ITEMS:
            cls.flush(st.cuda_stream)

    @staticmethod
    def _merge(q):
        """problems of one slice -> one problem over the concatenated rows (queue order kept by first occurrence; the merged problem overwrites /
        accumulates as its first member did, later members were accumulations by construction)"""
        seen = {}
        for i, t in enumerate(q):
            seen.setdefault((t[2].data_ptr(), t[3].data_ptr() if t[3] is not None else 0, t[0].dtype, t[0].shape[1], t[1].shape[1]), []).append(i)
        # one merged problem per SLICE: two keys on one weight slice (same arena pointer, another shape / bias slice / dtype: e.g. a
        # concatenated q|k|v sink and q alone on a shared module) would put an overwriting and an accumulating writer of the same words
        # into one launch — not reachable with today's models; refuse loudly rather than race
        ptrs = [k[0] for k in seen]
        if len(set(ptrs)) != len(ptrs):
            raise RuntimeError('WgradQueue: two queued weight-gradient problems write one arena slice with different shapes / bias slices')
        if len(seen) == len(q):
            return q
        out = []
        for key, idx in seen.items():
            t0 = q[idx[0]]
            if len(idx) == 1:
                out.append((idx[0], t0))
                continue
            dy = torch.cat([q[i][0] for i in idx], 0)
            x = torch.cat([q[i][1] for i in idx], 0)
            out.append((idx[0], (dy, x, t0[2], t0[3], t0[4])))
        out.sort(key=lambda e: e[0])
        return [t for _, t in out]

    @classmethod
    def arm(cls):
        """(inside a backward pass) have the autograd engine call _end_of_backward when this pass ends."""
        if not cls._callback_armed:
            cls._callback_armed = True
            torch.autograd.Variable._execution_engine.queue_callback(cls._end_of_backward)

    @classmethod
    def _end_of_backward(cls):
        cls._callback_armed = False
        cls.flush()
        Branch.join_all()           # grouped launches on side streams must land before the caller's stream goes on
        LnReduceQueue.flush()       # (after the join: the partials may have been produced on side streams)

    @classmethod
    def reset(cls):
        """Drop queued problems (GradArena.zero() calls this: anything still queued at the start of a step belongs to a
        backward pass that was aborted by an exception — its tensors must not be written into the new step)."""
        cls.queues, cls.pending_ids, cls._callback_armed = {}, {}, False
        cls.distinct = {}
        cls._balanced_i = 0
        LnReduceQueue.items = []

    @classmethod
    def flush_param(cls, param_id):
        """A queued write of this parameter's slice must land before the caller touches the slice on ITS stream."""
        h = cls.pending_ids.get(param_id)
        if h is not None:
            st = cls.queues[h][0]
            cls.flush(h)
            cur = torch.cuda.current_stream()
            if cur.cuda_stream != h:
                cur.wait_stream(st)

    @classmethod
    def flush(cls, handle=None):
        """Launch the queued problems of one stream (or of every stream), each group on its own stream."""
        for h in ([handle] if handle is not None else list(cls.queues)):
            ent = cls.queues.get(h)
            if not ent or not ent[1]:
                continue
            st, q = ent
            cls.queues[h] = (st, [])
            cls.distinct.pop(h, None)
            for pid in [k for k, v in cls.pending_ids.items() if v == h]:
                del cls.pending_ids[pid]
            with torch.cuda.stream(st):
                q = cls._merge(q)
                for i in range(0, len(q), cls.MAX):
                    cls._launch(q[i:i + cls.MAX])

    # contraction-balanced launch as a tuner candidate: opt-in.  Same-box A/B of the step with it among the candidates: 5.321 / 5.316 ms
    # without, 5.322 / 5.315 with (profiles/round5_wgrad_balanced.txt) — it wins only on groups that mix 8640-row and 3840-row problems.
    USE_BALANCED = os.environ.get('GOAT_WGRAD_BALANCED', '0') == '1'
    CANDIDATES = ((256, 3), (128, EIGHT_WAVES | 2), (tile(256, 256), 2)) + (      # tile configurations a group may run on
        ((tile(256, 256), PINGPONG | 2), (tile(128, 256), PINGPONG | 2), (256, PINGPONG | 2)) if USE_PP else ()) + (
        ((tile(256, 256), BALANCED | PINGPONG | 2),) if USE_PP and USE_BALANCED else ())
    _balanced_ws = {}       # (device, tile, i) -> zeroed workspace of the i-th balanced launch of a step (-1: the tuner's).  Launches of one step may
    _balanced_i = 0         # overlap on different streams, so each has its own; the eager warm-up step allocates them, the captured step finds them
    _last_ws = None         # (an allocation inside a capture would put its zero fill into the graph)

    @classmethod
    def _balanced_args(cls, cfg):
        """(workspace pointer, bytes) of the launch _run just made"""
        return (cls._last_ws.data_ptr(), cls._last_ws.numel())

    @classmethod
    def _run(cls, arr, n, cfg, tuning=False):
        """one grouped launch on the current stream -> status"""
        if not cfg[1] & BALANCED:
            return _lib.lib().goat_wgrad_grouped(_stream(), ctypes.addressof(arr), n, cfg[0], cfg[1])
        i = -1
        if not tuning:
            i, cls._balanced_i = cls._balanced_i, cls._balanced_i + 1
        key = (torch.cuda.current_device(), cfg[0], i)
        ws = cls._balanced_ws.get(key)
        if ws is None:
            nb = _lib.lib().goat_wgrad_balanced_ws_bytes(cfg[0])
            if nb <= 0:
                return -1
            ws = cls._balanced_ws[key] = torch.zeros(nb, dtype=torch.uint8, device='cuda')
        cls._last_ws = ws
        return _lib.lib().goat_wgrad_grouped_balanced(_stream(), ctypes.addressof(arr), n, cfg[0], ws.data_ptr(), ws.numel())

    tuned = {}              # group signature (rows, n_out, n_in per problem) -> configuration, timed on first sight (tuning.AUTOTUNE)
    TUNE = os.environ.get('GOAT_WGRAD_GROUP_TUNE', '1') != '0'
    FORCE_TUNE = False      # time unseen groups even while tuning.AUTOTUNE is off (bench.py keeps GEMM-shape tuning off around its T = 15 rollout graphs: ~100 shapes; a group costs milliseconds)

    @staticmethod
    def _fill(arr, items, scratch=None):
        for i, (dy, x, w, b, acc) in enumerate(items):
            p = arr[i]
            p.dy, p.ld_dy, p.x, p.ld_x = _ptr(dy), dy.stride(0), _ptr(x), x.stride(0)
            if scratch is None:
                p.dw, p.ld_dw, p.dbias, p.accumulate = _ptr(w), w.stride(0), (_ptr(b) if b is not None else None), acc
            else:
                p.dw, p.ld_dw, p.dbias, p.accumulate = _ptr(scratch[i]), scratch[i].stride(0), None, 0
            p.rows, p.n_out, p.n_in = dy.shape[0], dy.shape[1], x.shape[1]

    @classmethod
    def _tail_split(cls, q, rows=256, cols=128):
        """indices of the problems to run in a second launch on half-size tiles, or None.  With rows x cols tiles a group is a whole
        number of rounds over the 256 CUs plus a tail (e.g. sixteen text-layer problems on 256 x 128: 864 tiles = 3.375 rounds, the last
        one 37 % full; six text layers on 256 x 256: 648 tiles = 2.53 rounds).  Problems whose tiles add up to just over the tail are
        taken out and run afterwards on tiles of half the size (half the duration): e.g. 2 full rounds + 1.06 half rounds instead of 3."""
        if len(q) < 2 or len({t[0].shape[0] for t in q}) != 1:        # tiles of equal duration only (same contraction length)
            return None
        tiles = [((t[0].shape[1] + rows - 1) // rows) * ((t[1].shape[1] + cols - 1) // cols) for t in q]
        total, ncu = sum(tiles), n_cu()
        rem = total % ncu
        if total < ncu or rem == 0 or rem > ncu * 13 // 16:
            return None
        best = None                                                   # smallest subset sum >= rem (n <= 24: dynamic programme over sums)
        reach = {0: ()}
        for i, t in enumerate(tiles):
            for sm, idx in list(reach.items()):
                if sm + t not in reach:
                    reach[sm + t] = idx + (i,)
        for sm in sorted(reach):
            if sm >= rem:
                best = reach[sm]
                break
        if not best or len(best) == len(q):
            return None
        return best

    @classmethod
    def _plans(cls, q):
        n = len(q)
        plans = [[(tuple(range(n)), c)] for c in cls.CANDIDATES]
        tail = cls._tail_split(q)
        if tail is not None:
            head = tuple(i for i in range(n) if i not in tail)
            plans.append([(head, (256, 3)), (tail, (128, EIGHT_WAVES | 2))])
        if USE_PP:          # round 5: the same cut for the ping-pong 256 x 256 tile (tail on 128 x 256: half the rows, same columns)
            tail = cls._tail_split(q, 256, 256)
            if tail is not None:
                head = tuple(i for i in range(n) if i not in tail)
                plans.append([(head, (tile(256, 256), PINGPONG | 2)), (tail, (tile(128, 256), PINGPONG | 2))])
        return plans

    @classmethod
    def _pick_plan(cls, q):
        """[(problem indices, tile configuration)]: the launches this group runs as.  Which plan is fastest depends on the mix of
        problem sizes (whole rounds of tiles over the 256 CUs; e.g. eight text-layer problems: 787 TFLOP/s on 128x128 / 8 waves against
        720 on 256x128), so each distinct group is timed once on scratch outputs, cold caches, when autotuning is on and no graph is
        being captured."""
        n = len(q)
        default = [(tuple(range(n)), cls.cfg)]
        if 'GOAT_WGRAD_GROUP_CFG' in os.environ or not cls.TUNE:
            return default
        key = tuple((t[0].shape[0], t[0].shape[1], t[1].shape[1]) for t in q)
        plan = cls.tuned.get(key)
        log = os.environ.get('GOAT_WGRAD_PLAN_LOG')
        if plan is not None:
            tuning.STATS['wgrad_tuned'] += 1
            if log == '2':
                import sys
                print('[wgrad group] hit  %d problems rows %s capturing=%s -> %s' % (n, sorted({t_[0].shape[0] for t_ in q}), torch.cuda.is_current_stream_capturing(),
                      ' + '.join('%d x %s %s' % (len(i_), tile_name(c_[0]), stage_name(c_[1])) for i_, c_ in plan)), file=sys.stderr)
            return plan
        if not (tuning.AUTOTUNE or cls.FORCE_TUNE) or tuning.PROFILE is not None or torch.cuda.is_current_stream_capturing():
            tuning.STATS['wgrad_default'] += 1           # a group nobody timed runs the default tile (round 5: 10 of the 12 groups of the captured headline cycle did)
            if len(tuning.STATS_LOG) < 256:
                tuning.STATS_LOG.append(('wgrad', key, bool(torch.cuda.is_current_stream_capturing())))
            if log == '2':
                import sys
                print('[wgrad group] MISS %d problems rows %s autotune=%s profile=%s capturing=%s -> default' % (
                    n, sorted({t_[0].shape[0] for t_ in q}), tuning.AUTOTUNE, tuning.PROFILE is not None, torch.cuda.is_current_stream_capturing()), file=sys.stderr)
            return default
        scratch = [torch.empty((t[0].shape[1], t[1].shape[1]), dtype=torch.float32, device=t[0].device) for t in q]
        best = None
        for cand in cls._plans(q):
            parts = []
            for idx, cfg in cand:
                arr = (_lib.WgradProblem * len(idx))()
                cls._fill(arr, [q[i] for i in idx], [scratch[i] for i in idx])
                parts.append((arr, len(idx), cfg))

            def run():
                for arr, m, cfg in parts:
                    _lib.check(cls._run(arr, m, cfg, tuning=True), 'goat_wgrad_grouped (tuning)')
            try:
                t = _time_cfg(run, reps=int(os.environ.get('GOAT_WGRAD_TUNE_REPS', '9')))      # (a group runs 0.1-0.5 ms: nine repetitions cost nothing and the picks stop flipping between runs)
            except RuntimeError:
                continue
            if os.environ.get('GOAT_WGRAD_PLAN_LOG'):
                import sys
                print('[wgrad group] %d problems rows %s: %s -> %.1f us' % (n, sorted({t_[0].shape[0] for t_ in q}), ' + '.join(
                    '%d x %s %s' % (len(i_), tile_name(c_[0]), stage_name(c_[1])) for i_, c_ in cand), t * 1e3), file=sys.stderr)
            if best is None or t < best[0]:
                best = (t, cand)
        plan = cls.tuned[key] = best[1] if best is not None else default
        return plan

    ORDER = os.environ.get('GOAT_WGRAD_ORDER', 'spread')       # 'queue': the order the backward pass produced the problems in

    @classmethod
    def _spread(cls, q):
        """The problems of a group re-ordered so that every contraction length (rows) is spread evenly over the sequence.  The group
        kernel gives each XCD one contiguous chunk of the tile order (shared operand panels stay in one L2); a tile's duration is
        proportional to its contraction length, so a queue that holds the 8640-row panorama problems in one run and the 3840-row text
        problems in another hands some XCDs 2.25 x the work of others.  Problems write distinct slices: any order is valid."""
        if cls.ORDER != 'spread' or len({t[0].shape[0] for t in q}) < 2:
            return q
        by = {}
        for i, t in enumerate(q):
            by.setdefault(t[0].shape[0], []).append(i)
        n = len(q)
        slots = sorted(((k + 0.5) * n / len(ix), -rows, i) for rows, ix in by.items() for k, i in enumerate(ix))
        return [q[i] for _, _, i in slots]

    @classmethod
    def _launch(cls, q):
        q = cls._spread(q)
        for idx, cfg in cls._pick_plan(q):
            items = [q[i] for i in idx]
            n = len(items)
            arr = (_lib.WgradProblem * n)()
            cls._fill(arr, items)
            if tuning.PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            st = cls._run(arr, n, cfg)
            if tuning.PROFILE is not None:
                e1.record()
                fl = sum(2.0 * t[0].shape[0] * t[0].shape[1] * t[1].shape[1] for t in items)
                by = sum((t[0].shape[0] * t[0].shape[1] + t[1].shape[0] * t[1].shape[1]) * 2 + t[0].shape[1] * t[1].shape[1] * 4 for t in items)
                tuning.PROFILE.append((e0, e1, fl, ('grouped wgrad', n, by, 0, 1, 'v2 t11 %s %s' % (tile_name(cfg[0]), stage_name(cfg[1]))),
                                (('goat_wgrad_grouped', (ctypes.addressof(arr), n, cfg[0], cfg[1]), (arr, items)) if not cfg[1] & BALANCED else
                                 ('goat_wgrad_grouped_balanced', (ctypes.addressof(arr), n, cfg[0]) + cls._balanced_args(cfg), (arr, items)))))
            _lib.check(st, 'goat_wgrad_grouped(n=%d)' % n)


class LnReduceQueue:
    """Deferred dgamma / dbeta of LayerNorm backward.  With a gradient arena attached the two vectors are not needed until
    the backward pass ends; goat_ln_bwd then only leaves its per-block column partials behind (accumulate = 2) and ONE
    goat_ln_reduce_batched launch per backward pass adds the partials of every LayerNorm call to the arena slices — instead
    of ~1500 contended float atomics per block, or a reduction launch per call (scripts/ln_bench.py: 15.8 -> 9-10 us per call at
    3840 rows).  Deterministic.  Flushed by WgradQueue's end-of-backward callback."""
    enabled = os.environ.get('GOAT_LN_DEFER', '1') != '0'
    MIN_ROWS = 64
    items = []              # (ws, dgamma sink, dbeta sink, nparts, H): tensors kept alive until the launch

    @classmethod
    def push(cls, ws, dg, db, nparts, H):
        cls.items.append((ws, dg, db, nparts, H))
        WgradQueue.arm()

    @classmethod
    def flush(cls):
        items, cls.items = cls.items, []
        by_h = {}
        for it in items:
            by_h.setdefault(it[4], []).append(it)
        for H, group in by_h.items():
            arr = (_lib.LnPartial * len(group))()
            for e, (ws, dg, db, nparts, _) in zip(arr, group):
                e.ws, e.dgamma, e.dbeta, e.nparts = _ptr(ws), _ptr(dg), _ptr(db), nparts
            _lib.check(_lib.lib().goat_ln_reduce_batched(_stream(), ctypes.addressof(arr), len(group), H), 'goat_ln_reduce_batched')

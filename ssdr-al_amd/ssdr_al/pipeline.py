"""The hot path end to end, device-resident: grid-subsample -> tile -> KNN pyramid -> RandLA-Net inference ->
uncertainty / region statistics -> candidate features -> per-cloud chamfer graph + propagation -> FPS selection.

This is the sequencing the reference spreads over data_prepare_s3dis.py:58, s3dis_dataset.py:115-183,
sampler2.py:580-642 (prediction) and :736-781 (the gcn_fps branch of sampling); files, pickles and the label
simulation are out of scope.  All arithmetic happens in libssdr_al.so; the host only moves small index lists."""
import ctypes as C
import os
import time

import numpy as np

from . import _lib, randlanet
from ._lib import DevArray
from .helper_tool import ConfigS3DIS


class HotPath:
    def __init__(self, weights, config=ConfigS3DIS, sampler_args=("sb", "WetSU", "clsbal", "gcn_fps"), gcn_number=1, gcn_top=0,
                 select_per_tile=37, labeled_per_tile=15, seed=0, precision="f32"):
        self.cfg = config
        self.net = randlanet.Network(config).load(weights).set_precision(precision)
        self.sampler_args = list(sampler_args)
        self.gcn_number, self.gcn_top = gcn_number, gcn_top
        self.select_per_tile, self.labeled_per_tile = select_per_tile, labeled_per_tile
        self.seed = seed
        self.stream = None          # stream of the pyramid .. scoring stages (None = the library's main stream)
        self.front_stream = None    # stream of the front end (grid-subsample + tiles)
        self.knn_stream = None      # stream of the KNN pyramid (None = self.stream)
        self.score_stream = None    # stream of the scoring stage (None = self.stream)
        self.global_order = None
        self.rooms = []
        self.timing = None

    # ---- setup (untimed): upload raw rooms, fix the per-room randomness, derive superpoints -------------------
    def load_rooms(self, rooms, room_ids=None):
        """room_ids: global ids of the rooms (sharded runs); all host-drawn randomness is a function of (seed, id), so a
        sharded run and a single-process run over the union see the same tiles."""
        cfg = self.cfg
        N = cfg.num_points
        self.B = len(rooms)
        self.room_ids = list(range(len(rooms))) if room_ids is None else list(room_ids)
        self.rooms = []
        centers, perms, dups = [], [], []
        for (xyz, rgb, lab), rid in zip(rooms, self.room_ids):
            self.rng = np.random.default_rng([self.seed, rid])
            n = len(xyz)
            pick = xyz[self.rng.integers(0, n)] + self.rng.normal(0, cfg.noise_init / 10, 3)      # s3dis_dataset.py:119-126
            r = dict(n=n, center=np.ascontiguousarray(pick, np.float32),
                     perm=self.rng.permutation(N).astype(np.int32),                             # DP.shuffle_idx :137
                     dup=self.rng.random(N).astype(np.float32))                                 # DP.data_aug's np.random.choice
            centers.append(r["center"]); perms.append(r["perm"]); dups.append(r["dup"])
            self.rooms.append(r)
        # the rooms of the batch live concatenated in HBM: one batched call per front-end stage
        self.room_off = np.concatenate([[0], np.cumsum([len(r[0]) for r in rooms])]).astype(np.int64)
        nt = int(self.room_off[-1])
        self.raw_p = DevArray.from_host(np.concatenate([r[0] for r in rooms]).astype(np.float32))
        self.raw_c = DevArray.from_host(np.concatenate([r[1] for r in rooms]).astype(np.float32))
        self.raw_l = DevArray.from_host(np.concatenate([r[2] for r in rooms]).astype(np.int32).reshape(-1, 1))
        self.sub_p = DevArray((nt, 3), np.float32); self.sub_c = DevArray((nt, 3), np.float32); self.sub_l = DevArray((nt, 1), np.int32)
        self.sub_m = DevArray((len(rooms) + 1,), np.int64)
        self.centers = np.ascontiguousarray(np.stack(centers), np.float32)
        self.perm = DevArray.from_host(np.stack(perms)); self.dup = DevArray.from_host(np.stack(dups))
        B = self.B
        self.xyz = DevArray((B, N, 3), np.float32); self.feat = DevArray((B, N, 6), np.float32)
        L, K = cfg.num_layers, cfg.k_n
        self.sizes = [N]
        for ratio in cfg.sub_sampling_ratio:
            self.sizes.append(self.sizes[-1] // ratio)
        self.neigh = [DevArray((B, self.sizes[i], K), np.int32) for i in range(L)]
        self.interp = [DevArray((B, self.sizes[i], 1), np.int32) for i in range(L)]
        self.probs = DevArray((B * N, cfg.num_classes), np.float32); self.f32 = DevArray((B * N, 32), np.float32)
        self.unc = DevArray((B * N,), np.float32); self.cls = DevArray((B * N,), np.int32)
        # superpoints of the (fixed) tiles: computed once from a dry run of the geometric front end
        self._front_end()
        _lib.sync()
        from .synthetic import superpoints_from_tile
        tiles = self.xyz.to_host()
        offs, pts, cloud = [np.zeros(1, np.int32)], [], []
        for b in range(B):
            o, p = superpoints_from_tile(tiles[b])
            pts.append(p + b * N); offs.append(o[1:] + offs[-1][-1]); cloud.append(np.full(len(o) - 1, b, np.int32))
        self.sp_off_h = np.concatenate(offs).astype(np.int32); self.sp_pts_h = np.concatenate(pts).astype(np.int32)
        self.sp_cloud_h = np.concatenate(cloud)
        self.S = len(self.sp_off_h) - 1
        self.sp_base = [int(np.flatnonzero(self.sp_cloud_h == b)[0]) for b in range(B)]
        self.sp_off = DevArray.from_host(self.sp_off_h); self.sp_pts = DevArray.from_host(self.sp_pts_h)
        self.region_unc = DevArray((self.S,), np.float64); self.dom = DevArray((self.S,), np.int32); self.dom_cnt = DevArray((self.S,), np.int32)
        self.sorted_inds = DevArray((self.S,), np.int32)
        # "already labelled" superpoints (stand-in for total_obj / labeled_region_reference_dict) and class list
        self.labeled = {}
        for b in range(B):
            ids = np.flatnonzero(self.sp_cloud_h == b)
            rng = np.random.default_rng([self.seed, self.room_ids[b], 1])
            self.labeled[b] = set(rng.choice(ids, min(self.labeled_per_tile, len(ids)), replace=False).tolist())
        self.selected_class_list = DevArray.from_host(np.random.default_rng([self.seed, 999983]).integers(0, cfg.num_classes, 4000).astype(np.int32))
        self.hist = DevArray((64,), np.int32)
        self.labeled_mask = np.zeros(self.S, bool)
        for b in self.labeled:
            self.labeled_mask[list(self.labeled[b])] = True
        self.sp_size_h = np.diff(self.sp_off_h)
        return self

    # ---- stages ------------------------------------------------------------------------------------------------
    def _front_end(self):
        """grid-subsample + tile of every room of the batch, one batched launch sequence per stage (on front_stream)."""
        cfg, L = self.cfg, _lib.lib()
        st = self.front_stream
        _lib.check(L.ssdr_grid_subsample_batch_dev(self.raw_p.ptr, self.raw_c.ptr, 3, self.raw_l.ptr, 1, _lib.ptr(self.room_off), self.B, cfg.sub_grid_size,
                                                   self.sub_p.ptr, self.sub_c.ptr, self.sub_l.ptr, self.sub_m.ptr, st))
        _lib.check(L.ssdr_tile_select_batch_dev(self.sub_p.ptr, self.sub_c.ptr, 3, self.sub_m.ptr, _lib.ptr(self.room_off), self.B, _lib.ptr(self.centers),
                                                cfg.num_points, self.perm.ptr, self.dup.ptr, 1.0 / 255.0, self.xyz.ptr, self.feat.ptr, None, st))

    def _pyramid(self):
        cfg = self.cfg
        arr = C.c_void_p * cfg.num_layers
        r = np.asarray(cfg.sub_sampling_ratio, np.int32)
        _lib.check(_lib.lib().ssdr_knn_pyramid_dev(self.xyz.ptr, self.B, cfg.num_points, cfg.num_layers, _lib.ptr(r), cfg.k_n,
                                                   arr(*[a.ptr for a in self.neigh]), None, arr(*[a.ptr for a in self.interp]),
                                                   self.knn_stream if self.knn_stream is not None else self.stream))

    def _infer(self):
        self.net.infer_dev(self.B, self.cfg.num_points, self.feat.ptr, self.xyz.ptr, [a.ptr for a in self.neigh], [a.ptr for a in self.interp],
                           self.probs.ptr, self.f32.ptr, self.stream)

    def _score(self, comm=None):
        self._score_async(comm)
        self._score_finish(comm)

    def _score_async(self, comm=None):
        """device part of the scoring: enqueued, never waits (with a communicator: up to the local class histogram)"""
        cfg, L = self.cfg, _lib.lib()
        n = self.B * cfg.num_points
        um = {"lc": 0, "entropy": 1, "sb": 2}[[a for a in self.sampler_args if a in ("lc", "entropy", "sb")][0]]
        rm = {"mean": 0, "sum_weight": 1, "WetSU": 2}[[a for a in self.sampler_args if a in ("mean", "sum_weight", "WetSU")][0]]
        st = self.score_stream if self.score_stream is not None else self.stream
        _lib.check(L.ssdr_point_uncertainty_dev(self.probs.ptr, n, cfg.num_classes, um, self.unc.ptr, self.cls.ptr, st))
        _lib.check(L.ssdr_region_stats_dev(self.unc.ptr, self.cls.ptr, self.sp_off.ptr, self.sp_pts.ptr, self.S, cfg.num_classes, rm,
                                           self.region_unc.ptr, self.dom.ptr, self.dom_cnt.ptr, st))
        nsel = self.selected_class_list.shape[0]
        if "clsbal" in self.sampler_args:
            if comm is None:
                _lib.check(L.ssdr_clsbal_dev(self.dom.ptr, self.S, self.selected_class_list.ptr, nsel, self.region_unc.ptr, st))
            else:       # the already-selected list is counted once, on rank 0
                _lib.check(L.ssdr_class_hist_dev(self.dom.ptr, self.S, self.selected_class_list.ptr, nsel if comm.rank == 0 else 0, self.hist.ptr, st))
        if comm is None:
            _lib.check(L.ssdr_rank_regions_dev(self.region_unc.ptr, self.S, self.sorted_inds.ptr, st))
            self.global_order = None

    def _score_finish(self, comm=None):
        """host-synchronous part (communicator only): exchanges 1 and 2, then the global ranking"""
        if comm is None:
            return
        L = _lib.lib()
        st = self.score_stream if self.score_stream is not None else self.stream
        nsel = self.selected_class_list.shape[0]
        if "clsbal" in self.sampler_args:       # exchange 1: the class histogram is global
            _lib.sync(st)
            h = comm.allreduce_sum(np.concatenate([self.hist.to_host().astype(np.int64), [self.S]]))
            self.hist = DevArray.from_host(h[:64].astype(np.int32))
            _lib.check(L.ssdr_clsbal_hist_dev(self.dom.ptr, self.S, self.hist.ptr, int(h[64]) + nsel, self.region_unc.ptr, st))
        # exchange 2: rank the regions of ALL ranks; labelled regions are taken out before the cut
        _lib.sync(st)
        u = self.region_unc.to_host()
        lab = np.zeros(self.S, bool)
        for b in self.labeled:
            lab[list(self.labeled[b])] = True
        allu, counts = comm.allgather_var(np.where(lab, -np.inf, u))
        d_all = DevArray.from_host(allu); d_ord = DevArray((len(allu),), np.int32)
        _lib.check(L.ssdr_rank_regions_dev(d_all.ptr, len(allu), d_ord.ptr, st))
        _lib.sync(st)
        base = int(sum(counts[: comm.rank]))
        self.global_order = (d_ord.to_host(), allu, base, self.select_per_tile * self.B * comm.world)

    def _candidates(self, sorted_inds):
        """create_file_top_and_all + the candidate rule of sampling() (sampler2.py:533-552, :745-753) on index lists.
        Candidate order is canonical (cloud ascending, then descending uncertainty): the reference's own order depends
        on a shuffled DataLoader (sampler2.py:323) and carries no meaning."""
        if self.global_order is None:
            order = np.asarray(sorted_inds, np.int64)
            keep = ~self.labeled_mask[order]                 # labelled regions never compete
            cand = order[keep]
            batch_size = min(self.select_per_tile * self.B, len(order))
            local = cand                                      # all candidates are local
            in_top = np.arange(len(cand)) < batch_size
        else:
            order, allu, base, batch_size = self.global_order
            order = np.asarray(order, np.int64)
            cand_g = order[allu[order] != -np.inf]           # labelled regions were masked to -inf and sort last
            in_top_g = np.arange(len(cand_g)) < batch_size
            mine = (cand_g >= base) & (cand_g < base + self.S)
            local, in_top = cand_g[mine] - base, in_top_g[mine]
        cloud = self.sp_cloud_h[local]
        grp = np.argsort(cloud, kind="stable")               # cloud ascending, descending uncertainty inside a cloud
        local, in_top, cloud = local[grp], in_top[grp], cloud[grp]
        ntop = np.bincount(cloud[in_top], minlength=self.B)  # selected_num per cloud (len(file_list_top[cloud]))
        first = np.searchsorted(cloud, np.arange(self.B))
        pos = np.arange(len(local)) - first[cloud]
        take = pos < 2 * ntop[cloud]                          # candidates = first 2 x selected_num of the cloud (:748)
        unl = [(int(b), int(s)) for b, s in zip(cloud[take], local[take])]
        sampling_batch = int(ntop.sum())
        lab = [(b, s) for b in sorted(self.labeled) for s in sorted(self.labeled[b])]
        return unl, lab, sampling_batch

    def _select(self, comm=None):
        self._select_issue(comm)
        return self._select_collect()

    def _select_issue(self, comm=None):
        """everything of the selection up to the enqueued FPS chain (the host decisions and uploads happen here)"""
        L = _lib.lib()
        sorted_inds = self.sorted_inds.to_host() if self.global_order is None else None   # small D2H: the host decides the candidate lists
        unl, lab, sampling_batch = self._candidates(sorted_inds)
        refs = unl + lab
        sel = np.array([s for _, s in refs], np.int32)
        # every cloud's chamfer graph and propagation hop in one batched call (rows grouped cloud by cloud)
        ref_cloud = np.fromiter((c for c, _ in refs), np.int64, len(refs))
        order = np.argsort(ref_cloud, kind="stable").astype(np.int32)
        clouds, counts = np.unique(ref_cloud, return_counts=True)
        counts = counts.astype(np.int64)
        coff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        boff = np.concatenate([[0], np.cumsum(counts * counts)]).astype(np.int64)
        ntot, nmax, nsq = int(coff[-1]), int(counts.max()), int(boff[-1])
        # ONE upload for all the small index tables (each separate copy is a host round trip behind the kernels in flight)
        parts = [sel, sel[order], order, coff, boff.view(np.int32)]
        offs = np.cumsum([0] + [(len(p) + 3) // 4 * 4 for p in parts])           # 16-byte aligned pieces
        pack = np.zeros(offs[-1], np.int32)
        for p, o in zip(parts, offs):
            pack[o:o + len(p)] = p
        d_pack = DevArray.from_host(pack)
        d_sel, d_gsel, d_rows, d_coff, d_boff = (d_pack.ptr + 4 * int(o) for o in offs[:-1])
        d_mf = DevArray((len(sel), 32), np.float32)
        d_v = DevArray((len(sel), 32), np.float64); d_comb = DevArray((len(sel), 32), np.float64)
        d_tmp = [DevArray(d_v.shape, np.float64), DevArray(d_v.shape, np.float64)]
        _lib.check(L.ssdr_segment_mean_features_dev(self.f32.ptr, 32, self.cls.ptr, self.dom.ptr, self.sp_off.ptr, self.sp_pts.ptr, d_sel, len(sel), d_mf.ptr, None))
        # float32 -> float64 as np.concatenate / np.matmul promote it (V and the running sum comb start as the same values)
        _lib.check(L.ssdr_widen_f32_f64_dev(d_mf.ptr, len(sel) * 32, d_v.ptr, d_comb.ptr, None))
        d_cen = DevArray((ntot, 3), np.float64); d_dir = DevArray((nsq,), np.float64); d_adj = DevArray((nsq,), np.float64)
        _lib.check(L.ssdr_cloud_graph_batch_dev(self.xyz.ptr, self.sp_off.ptr, self.sp_pts.ptr, d_gsel, d_coff, d_boff, len(clouds), ntot, nmax,
                                                int(self.gcn_top), d_cen.ptr, d_dir.ptr, d_adj.ptr, None))
        src = d_v
        for hop in range(int(self.gcn_number)):
            dst = d_tmp[hop & 1]
            _lib.check(L.ssdr_propagate_batch_dev(d_adj.ptr, d_coff, d_boff, len(clouds), nmax, d_rows, src.ptr, 32, dst.ptr, d_comb.ptr, None))
            src = dst
        blocks = (d_pack, d_cen, d_dir, d_adj)
        n_unl = len(unl)
        self.unl_cloud_ids = np.array([self.room_ids[b] for b, _ in unl], np.int64)
        self.unl_sp = np.array([s - self.sp_base[b] for b, s in unl], np.int64)       # superpoint index inside its room
        if comm is not None:                                 # exchange 3: the candidates' propagated features (+ their ids)
            _lib.sync()
            comb_all, _ = comm.allgather_var(d_comb.to_host()[:n_unl])
            ids_all, _ = comm.allgather_var(np.stack([self.unl_cloud_ids, self.unl_sp], 1))
            sampling_batch = int(comm.allreduce_sum(np.array([sampling_batch], np.int64))[0])
            d_comb = DevArray.from_host(comb_all); n_unl = len(comb_all)
            self.unl_cloud_ids, self.unl_sp = ids_all[:, 0], ids_all[:, 1]
            self.comb_all = comb_all
        d_out = DevArray((sampling_batch,), np.int32)
        start = 0                                            # np.random.randint(0, n) in the reference (:133); fixed here
        _lib.check(L.ssdr_fps_dev(d_comb.ptr, n_unl, 32, start, sampling_batch, d_out.ptr, None))
        self._keep = (blocks, d_v, d_tmp, d_mf, d_comb)
        self._pending = (d_out, unl)

    def _select_collect(self):
        """wait for the FPS chain of _select_issue and read the selection back"""
        d_out, unl = self._pending
        self._pending = None
        _lib.sync()
        sel = d_out.to_host()
        self.selected = [(int(self.unl_cloud_ids[i]), int(self.unl_sp[i])) for i in sel]      # (room id, superpoint in room)
        return sel, unl

    def step(self, comm=None, timed_stages=False):
        """One pass of the hot path over the loaded batch of rooms.  Returns the selected candidate indices."""
        t = [time.perf_counter()]

        def mark():
            if timed_stages:
                _lib.sync(); t.append(time.perf_counter())
        self._front_end(); mark()
        self._pyramid(); mark()
        self._infer(); mark()
        self._score(comm); mark()
        out = self._select(comm); mark()
        if timed_stages:
            self.timing = dict(zip(("subsample+tile", "knn_pyramid", "randla_infer", "score", "select"), np.diff(t) * 1e3))
        return out


class Pipelined:
    """Overlap consecutive batches on separate buffer sets and HIP streams (software pipeline over the stages
    front end | KNN pyramid | network | scoring | selection).  `depth` batches are in flight:
    depth 2: selection of batch k (latency-bound: host decisions, one workgroup of FPS) on the main stream next to
             everything else of batch k+1 on a second stream;
    depth 3: front end (subsample + tiles) on its own stream, one batch further ahead;
    depth 4: front end + KNN pyramid | network | scoring | selection — three created streams + the main one; the choice when a
             framework's own streams (RCCL exchanges) share the process and its 4 hardware queues (92 vs 63 Mpoints/s);
    depth 5 (default): every stage on its own stream (one GPU, no framework streams: 99 vs 86 Mpoints/s for depth 4).
    Every batch still goes through every stage; `run(K)` finishes K selections."""
    STAGES = ("front", "knn", "infer", "score")
    GROUPS = {2: (0, 0, 0, 0), 3: (0, 1, 1, 1), 4: (0, 0, 1, 2), 5: (0, 1, 2, 3)}     # stage -> stream group

    def __init__(self, make_hot_path, depth=5, groups=None):
        """groups: optional stage -> stream-group tuple for (front, knn, infer, score), non-decreasing from 0; depth = last group + 2"""
        if groups is not None:
            depth = groups[-1] + 2
        assert groups is not None or depth in self.GROUPS
        L = _lib.lib()
        self.depth = depth
        self.group = dict(zip(self.STAGES, groups if groups is not None else self.GROUPS[depth]))
        # The first stream created after the library's own shares its hardware queue on this runtime (measured on
        # MI355X / ROCm 7.2: whatever stage sat on it serialised with the selection kernels of the main stream, 67 vs
        # 81 Mpoints/s at depth 4, for any GPU_MAX_HW_QUEUES): leave that one unused.
        self._spare = C.c_void_p()
        _lib.check(L.ssdr_stream_create(C.byref(self._spare)))
        self.streams = []
        for _ in range(depth - 1):
            st = C.c_void_p()
            _lib.check(L.ssdr_stream_create(C.byref(st)))
            self.streams.append(st.value)
        self.lead = {n: depth - 1 - g for n, g in self.group.items()}      # batches ahead of the selection
        self.hp = [make_hot_path() for _ in range(depth)]
        for h in self.hp:
            h.front_stream, h.knn_stream = self.streams[self.group["front"]], self.streams[self.group["knn"]]
            h.stream, h.score_stream = self.streams[self.group["infer"]], self.streams[self.group["score"]]
        self._drain()

    def _drain(self):
        _lib.sync()
        for st in self.streams:
            _lib.sync(st)

    def _stage(self, name, b):
        """enqueue one stage of batch b; a consumer stream first waits for everything its producer stream holds so far"""
        h = self.hp[b % self.depth]
        i = self.STAGES.index(name)
        if i > 0 and self.group[name] != self.group[self.STAGES[i - 1]]:
            _lib.check(_lib.lib().ssdr_stream_wait(self.streams[self.group[name]], self.streams[self.group[self.STAGES[i - 1]]]))
        if name == "score":
            h._score_async(self.comm)
        else:
            {"front": h._front_end, "knn": h._pyramid, "infer": h._infer}[name]()

    def run(self, steps, comm=None):
        self.comm = comm
        L = _lib.lib()
        out = None
        lead, first = self.lead, self.lead["front"]
        for b in range(min(steps, first)):                   # prologue: fill the pipe
            for name in self.STAGES:
                if b < lead[name]:
                    self._stage(name, b)
                    if name == "score":
                        self.hp[b % self.depth]._score_finish(comm)
        for k in range(steps):
            _lib.check(L.ssdr_stream_wait(None, self.streams[self.group["score"]]))    # main stream: batch k's scores are ready
            hk = self.hp[k % self.depth]
            hk._select_issue(comm)                           # batch k: host decisions + the whole selection chain enqueued ...
            late = []
            for b in range(k + 1, min(steps, k + first + 1)):   # ... the other stages are issued while its FPS chain (one workgroup,
                for name in self.STAGES:                        # ~1.6 ms) runs, instead of before it ...
                    if b == k + lead[name]:
                        self._stage(name, b)                 # the buffer set of batch b was last read by select(b - depth), done
                        if name == "score":
                            late.append(b)
            out = hk._select_collect()                       # ... and only then the host waits for the selection
            for b in late:                                   # with a communicator: exchanges 1 + 2 of batch k+1, host-synchronous, after the
                self.hp[b % self.depth]._score_finish(comm)  # selection, by when that batch's scoring kernels have long finished
        self._drain()
        return out

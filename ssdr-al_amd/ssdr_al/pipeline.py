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


class _Pairs:
    """A read-only sequence of (a[i], b[i]) integer pairs that behaves like the list of tuples it stands for (indexing, slicing, iteration, len, ==) and
    builds that list only when asked to: an AL round hands back 20 000 candidates and 10 000 picks, and building 30 000 Python tuples nobody may look at
    cost the collect 1.5 of its 2.2 ms."""
    __slots__ = ("a", "b", "_list")

    def __init__(self, a, b):
        self.a, self.b, self._list = np.asarray(a), np.asarray(b), None

    def tolist(self):
        if self._list is None:
            self._list = list(zip(self.a.tolist(), self.b.tolist()))
        return self._list

    def __len__(self):
        return len(self.a)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.tolist()[i]
        return (int(self.a[i]), int(self.b[i]))

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        return self.tolist() == (other.tolist() if isinstance(other, _Pairs) else other)

    def __repr__(self):
        return repr(self.tolist())


class HotPath:
    def __init__(self, weights, config=ConfigS3DIS, sampler_args=("sb", "WetSU", "clsbal", "gcn_fps"), gcn_number=1, gcn_top=0,
                 select_per_tile=37, labeled_per_tile=15, seed=0, precision="f32", selector="fps", tiles32=True, min_size=1, round_num=5,
                 label_seed=None, batch_size=None, max_size=None, chamfer_mode="f64"):
        self.cfg = config
        self.net = None if weights is None else randlanet.Network(config).load(weights).set_precision(precision).set_formulation(tiles32)
        self.sampler_args = list(sampler_args)
        self.gcn_number, self.gcn_top = gcn_number, gcn_top
        self.select_per_tile, self.labeled_per_tile = select_per_tile, labeled_per_tile
        self.batch_size = None if batch_size is None else int(batch_size)      # sampling()'s batch_size outright (default: select_per_tile x clouds)
        self.seed = seed
        # min_size: regions of fewer points are neither ranked nor used as labelled rows (sampler2.py:616, :628; the reference's default is 1);
        # round_num: the class-balanced draw of the labelled rows takes (round_num - 1) * 1000 of the labelled regions (sampler2.py:297-302; SURVEY
        # section 8d quotes the workload at round 5); label_seed seeds NumPy's legacy generator for that draw (the reference draws from np.random)
        # max_size: the Semantic3D flavour also drops regions of more than 1000 points from both populations (SSRD_AL_semantic3d/sampler2.py:644, :655)
        self.max_size = None if max_size is None else int(max_size)
        # chamfer_mode: "f64" (S3DIS, fps_gcn_cpu.create_cd) or "f32_cuda" (the Semantic3D code's create_cd_cuda values, fps_gcn_cuda.py:13-30)
        self.chamfer_mode = {"f64": 0, "f32_cuda": 1}[chamfer_mode]
        self.min_size, self.round_num = int(min_size), int(round_num)
        self.label_seed = int(seed if label_seed is None else label_seed)
        # "fps": farthest_features_sample over the candidates' propagated features (the gcn_fps branch, sampler2.py:736-781);
        # "kcenter": kCenterGreedy.select_batch_ over candidates + labelled regions with the labelled ones as already selected
        # (the step the reference's gcn branch ends with, gcn.py:247, kcenterGreedy.py:60-128; BASELINE configuration 4's global k-center)
        assert selector in ("fps", "kcenter")
        self.selector = selector
        self.fps_start = 0          # np.random.randint(0, n) in the reference (fps_gcn_cpu.py:133); fixed here
        self.stream = None          # stream of the pyramid .. scoring stages (None = the library's main stream)
        self.front_stream = None    # stream of the front end (grid-subsample + tiles)
        self.knn_stream = None      # stream of the KNN pyramid (None = self.stream)
        self.score_stream = None    # stream of the scoring stage (None = self.stream)
        self.sel_stream = None      # stream of the selection chain (None = the library stream)
        self.pipelined = False      # set by Pipelined: later batches are in flight on the stage streams when a selection is collected
        self.global_order = None
        self.rooms = []
        self.timing = None

    # ---- setup (untimed): upload raw rooms, fix the per-room randomness, derive superpoints -------------------
    def draw_room(self, xyz, rid):
        """the host-drawn randomness of one tile, a function of (seed, global room id): pick point, shuffle, padding draws"""
        cfg, N = self.cfg, self.cfg.num_points
        self.rng = np.random.default_rng([self.seed, rid])
        n = len(xyz)
        pick = xyz[self.rng.integers(0, n)] + self.rng.normal(0, cfg.noise_init / 10, 3)      # s3dis_dataset.py:119-126
        return dict(n=n, center=np.ascontiguousarray(pick, np.float32),
                    perm=self.rng.permutation(N).astype(np.int32),                             # DP.shuffle_idx :137
                    dup=self.rng.random(N).astype(np.float32))                                 # DP.data_aug's np.random.choice

    def load_rooms(self, rooms, room_ids=None):
        """room_ids: global ids of the rooms (sharded runs); all host-drawn randomness is a function of (seed, id), so a
        sharded run and a single-process run over the union see the same tiles."""
        cfg = self.cfg
        N = cfg.num_points
        self.B = len(rooms)
        self._dist = None           # the sharded run's static tables (labelled mask, sizes, exchange buffers) belong to the loaded rooms
        self.room_ids = list(range(len(rooms))) if room_ids is None else list(room_ids)
        self.rooms = []
        centers, perms, dups = [], [], []
        for (xyz, rgb, lab), rid in zip(rooms, self.room_ids):
            r = self.draw_room(xyz, rid)
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
        self.tile_l = DevArray((B * N,), np.int32)        # queried_pc_label (s3dis_dataset.py:141): the labelled regions' ground truth
        L, K = cfg.num_layers, cfg.k_n
        self.sizes = [N]
        for ratio in cfg.sub_sampling_ratio:
            self.sizes.append(self.sizes[-1] // ratio)
        self.neigh = [DevArray((B, self.sizes[i], K), np.int32) for i in range(L)]
        self.interp = [DevArray((B, self.sizes[i], 1), np.int32) for i in range(L)]
        self.probs = DevArray((B * N, cfg.num_classes), np.float32); self.f32 = DevArray((B * N, 32), np.float32)
        self.unc = DevArray((B * N,), np.float32); self.cls = DevArray((B * N,), np.int32)
        # superpoints of the (fixed) tiles: computed once from a dry run of the geometric front end.  The dry run also settles which
        # implementation of the batched grid subsample these rooms get: the bucket partition reports a cloud it cannot take (a grid of
        # more than 16384 buckets, a voxel of more than 1024 points) through its status, and the rooms then go through the sort
        self.subsample_method = 1 if os.environ.get("SSDR_SUBSAMPLE_METHOD") == "sort" else 0          # (development: A/B of the two implementations)
        self._front_end()
        _lib.sync()
        st = C.c_int32()
        rc = _lib.lib().ssdr_grid_subsample_status(self.front_stream, C.byref(st))
        if rc != 0 and (st.value & 6):
            self.subsample_method = 1
            self._front_end()
            _lib.sync()
            _lib.check(_lib.lib().ssdr_grid_subsample_status(self.front_stream, None))
        else:
            _lib.check(rc)
        from .synthetic import superpoints_from_tile
        tiles = self.xyz.to_host()
        offs, pts, cloud = [np.zeros(1, np.int32)], [], []
        for b in range(B):
            o, p = superpoints_from_tile(tiles[b])
            pts.append(p + b * N); offs.append(o[1:] + offs[-1][-1]); cloud.append(np.full(len(o) - 1, b, np.int32))
        labeled = {}
        sp_cloud = np.concatenate(cloud)
        for b in range(B):           # "already labelled" superpoints (stand-in for the regions total_obj["unlabeled"] no longer lists)
            ids = np.flatnonzero(sp_cloud == b)
            rng = np.random.default_rng([self.seed, self.room_ids[b], 1])
            labeled[b] = set(rng.choice(ids, min(self.labeled_per_tile, len(ids)), replace=False).tolist())
        self.n_pts = B * N
        self._set_regions(np.concatenate(offs), np.concatenate(pts), sp_cloud, labeled,
                          np.random.default_rng([self.seed, 999983]).integers(0, cfg.num_classes, 4000))
        return self

    @classmethod
    def from_clouds(cls, clouds, labeled, selected_class_list, config=ConfigS3DIS, room_ids=None, **kw):
        """The scoring + selection half alone, over clouds given with their network outputs (no front end, no network): clouds[b] = dict(xyz [n,3],
        gt [n], probs [n,C], feat [n,32], offsets [S_b+1], points) — what prediction() / compute_features see per cloud (sampler2.py:580-642,
        :313-342); labeled[b] = the regions of cloud b (ids inside the cloud) that total_obj["unlabeled"] no longer lists.  Clouds may differ in
        size: every array is the concatenation, superpoints are one CSR over it.  step_selection() runs the round."""
        n_of = [len(c["xyz"]) for c in clouds]
        p0 = np.concatenate([[0], np.cumsum(n_of)]).astype(np.int64)
        offs, pts, cloud, lab, s0 = [np.zeros(1, np.int64)], [], [], {}, 0
        for b, c in enumerate(clouds):
            o = np.asarray(c["offsets"], np.int64)
            offs.append(o[1:] + offs[-1][-1]); pts.append(np.asarray(c["points"], np.int64) + p0[b]); cloud.append(np.full(len(o) - 1, b, np.int32))
            lab[b] = set(int(s) + s0 for s in labeled[b]); s0 += len(o) - 1
        return cls.from_device(DevArray.from_host(np.concatenate([np.asarray(c["xyz"], np.float32) for c in clouds])),
                               DevArray.from_host(np.concatenate([np.asarray(c["probs"], np.float32) for c in clouds])),
                               DevArray.from_host(np.concatenate([np.asarray(c["feat"], np.float32) for c in clouds])),
                               DevArray.from_host(np.concatenate([np.asarray(c["gt"]).astype(np.int32) for c in clouds])),
                               np.concatenate(offs), np.concatenate(pts), np.concatenate(cloud), lab, selected_class_list, config, room_ids=room_ids, **kw)

    @classmethod
    def from_device(cls, xyz, probs, f32, labels, sp_off, sp_pts, sp_cloud, labeled, selected_class_list, config=ConfigS3DIS, room_ids=None, **kw):
        """from_clouds over arrays that are already resident (xyz [n,3] f32, probs [n,C] f32, f32 [n,32] f32, labels [n] i32: DevArray or anything
        with .ptr): the superpoints as one CSR over the n points (host arrays), sp_cloud [S] = the cloud of every superpoint (ascending),
        labeled[b] = GLOBAL ids of cloud b's labelled regions.  This is how one AL round ends at the reference's scale (ALRound): the network
        outputs of all clouds stay where the batches' inference left them."""
        hp = cls(None, config, **kw)
        sp_cloud = np.asarray(sp_cloud, np.int32)
        hp.B = int(sp_cloud.max()) + 1 if len(sp_cloud) else 0
        hp._dist = None; hp.room_ids = list(range(hp.B)) if room_ids is None else list(room_ids); hp.rooms = []      # room_ids: the clouds' global ids (sharded runs)
        hp.n_pts = int(xyz.shape[0])
        hp.xyz, hp.probs, hp.f32, hp.tile_l = xyz, probs, f32, labels
        hp.unc = DevArray((hp.n_pts,), np.float32); hp.cls = DevArray((hp.n_pts,), np.int32)
        hp._set_regions(sp_off, sp_pts, sp_cloud, labeled, selected_class_list)
        return hp

    def step_selection(self, comm=None):
        """scoring + selection over the resident network outputs (from_clouds); comm: the sharded run's exchanges"""
        self._score_async(comm)
        return self._select(comm)

    def _set_regions(self, sp_off, sp_pts, sp_cloud, labeled, selected_class_list):
        """superpoints (one CSR over the batch's points), which of them are already labelled, the already-selected class list; then everything
        the selection derives from them once: the ranked population, the labelled rows' draw, the device rule's static tables"""
        cfg, B = self.cfg, self.B
        self.sp_off_h = np.asarray(sp_off).astype(np.int32); self.sp_pts_h = np.asarray(sp_pts).astype(np.int32)
        self.sp_cloud_h = np.asarray(sp_cloud).astype(np.int32)
        self.S = len(self.sp_off_h) - 1
        self.sp_base = np.searchsorted(self.sp_cloud_h, np.arange(B)).astype(np.int64).tolist()      # (clouds are ascending: every cloud owns at least one region)
        self.sp_off = DevArray.from_host(self.sp_off_h); self.sp_pts = DevArray.from_host(self.sp_pts_h)
        self.region_unc = DevArray((self.S,), np.float64); self.dom = DevArray((self.S,), np.int32); self.dom_cnt = DevArray((self.S,), np.int32)
        self.gt_dom = DevArray((self.S,), np.int32); self.gt_purity = DevArray((self.S,), np.float64)
        self.sorted_inds = DevArray((self.S,), np.int32)
        self.selected_class_list = DevArray.from_host(np.asarray(selected_class_list).astype(np.int32).reshape(-1))
        self.hist = DevArray((64,), np.int32)
        self.sp_size_h = np.diff(self.sp_off_h)
        self.set_labeled(labeled)

    def set_labeled(self, labeled):
        """labeled[b] = the (global) ids of cloud b's regions that are already labelled (what total_obj["unlabeled"] no longer lists)"""
        cfg = self.cfg
        self._dist = None; self.global_order = None      # a sharded run's static tables (labelled mask, populations, the global draw) belong to the labelling
        self.labeled = labeled
        self.labeled_mask = np.zeros(self.S, bool)
        for b in self.labeled:
            self.labeled_mask[list(self.labeled[b])] = True
        # prediction()'s population (sampler2.py:612-631): unlabelled regions of at least min_size points are ranked; labelled ones of at least
        # min_size points are the pool the labelled rows are drawn from; the rest takes no part.  skip_mask = not in the ranked population
        big = self.sp_size_h >= self.min_size
        if self.max_size is not None:
            big &= self.sp_size_h <= self.max_size
        self.skip_mask = self.labeled_mask | ~big
        self.lab_pool = np.flatnonzero(self.labeled_mask & big)                 # cloud by cloud, ascending superpoint id
        # the pool's ground-truth dominant labels from the resident labels (the reference reads them from the cloud's PLY, :283-289)
        _lib.check(_lib.lib().ssdr_dominant_label_dev(self.tile_l.ptr, self.sp_off.ptr, self.sp_pts.ptr, self.S, max(cfg.num_classes, 1), self.gt_dom.ptr,
                                                      self.gt_purity.ptr, self.front_stream))
        _lib.sync(self.front_stream)
        self.lab_pool_dom = self.gt_dom.to_host()[self.lab_pool]
        self._draw_labelled(self.lab_pool, self.lab_pool_dom, np.arange(len(self.lab_pool)))

    def _draw_labelled(self, pool_sp, pool_dom_all, mine):
        """get_labeled_selection_cloudname_spidx_pointidx's draw (sampler2.py:294-302) over the pool given by its dominant labels `pool_dom_all` (all
        ranks' pools in cloud order for a sharded run; `mine` = this process's positions in it, pool_sp = their local superpoint ids)."""
        from . import sampler
        drawn = sampler.get_labeled_selection(pool_dom_all, self.cfg.num_classes, self.round_num, np.random.RandomState(self.label_seed))
        keep = np.zeros(len(pool_dom_all), bool); keep[drawn] = True
        rows = np.asarray(pool_sp, np.int64)[keep[np.asarray(mine, np.int64)]]
        self.lab_rows = {b: sorted(int(x) for x in rows[self.sp_cloud_h[rows] == b]) for b in range(self.B)}      # cloud ascending, superpoint ascending
        self._select_static()

    def _select_static(self):
        """Static tables and capacities of the device-side candidate rule (ssdr_gcn_fps_sampling_dev): which regions are labelled, where a
        cloud's superpoints start, its labelled regions, and upper bounds of what the rule can produce (the counts themselves are the
        ranking's and stay on the device)."""
        B, S = self.B, self.S
        base = np.concatenate([np.asarray(self.sp_base, np.int64), [S]])
        lab = [self.lab_rows.get(b, []) for b in range(B)]
        lab_off = np.concatenate([[0], np.cumsum([len(l) for l in lab])]).astype(np.int32)
        lab_sp = np.array([s for l in lab for s in l] + [0], np.int32)
        nvalid = np.array([int((~self.skip_mask[base[b]:base[b + 1]]).sum()) for b in range(B)], np.int64)
        nlab = np.diff(lab_off).astype(np.int64)
        batch = self.batch_size if self.batch_size is not None else self.select_per_tile * B      # sampling()'s batch_size
        picks = int(min(batch, nvalid.sum()))
        cap_unl = int(min(2 * picks, nvalid.sum()))
        share = np.minimum(2 * picks, nvalid)                    # a cloud offers at most 2 x (its top regions) <= 2 x picks, and what it has
        cap_rows = cap_unl + int(nlab.sum())
        # the largest sum of squared cloud shares: fill the roomiest clouds first
        left, sq = cap_unl, 0
        for i in np.argsort(-(share + nlab), kind="stable"):
            a = int(min(share[i], left)); left -= a
            sq += (a + int(nlab[i])) ** 2
        self._sel_static = dict(
            lab_cloud=np.repeat(np.arange(B, dtype=np.int64), nlab), lab_sp_h=lab_sp[:-1].astype(np.int64),
            d_lab=DevArray.from_host(self.skip_mask.astype(np.uint8)), d_base=DevArray.from_host(base.astype(np.int32)),
            d_lab_off=DevArray.from_host(lab_off), d_lab_sp=DevArray.from_host(lab_sp), n_lab=int(nlab.sum()), batch=batch, picks=picks, cap_unl=max(cap_unl, 1),
            cap_rows=max(cap_rows, 1), cap_nmax=max(int((share + nlab).max()) if B else 1, 1), cap_sq=max(int(sq), 1),
            d_result=DevArray((8 + picks + max(cap_rows, 1),), np.int32))

    # ---- stages ------------------------------------------------------------------------------------------------
    def _front_end(self):
        """grid-subsample + tile of every room of the batch, one batched launch sequence per stage (on front_stream)."""
        cfg, L = self.cfg, _lib.lib()
        st = self.front_stream
        _lib.check(L.ssdr_grid_subsample_set_method(self.subsample_method))
        _lib.check(L.ssdr_grid_subsample_batch_dev(self.raw_p.ptr, self.raw_c.ptr, 3, self.raw_l.ptr, 1, _lib.ptr(self.room_off), self.B, cfg.sub_grid_size,
                                                   self.sub_p.ptr, self.sub_c.ptr, self.sub_l.ptr, self.sub_m.ptr, st))
        _lib.check(L.ssdr_tile_select_batch_dev(self.sub_p.ptr, self.sub_c.ptr, 3, self.sub_m.ptr, _lib.ptr(self.room_off), self.B, _lib.ptr(self.centers),
                                                cfg.num_points, self.perm.ptr, self.dup.ptr, 1.0 / 255.0, self.xyz.ptr, self.feat.ptr, None, self.sub_l.ptr, self.tile_l.ptr, st))

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

    def _dist_setup(self, comm):
        """Static tables of the sharded run, exchanged once (untimed): every rank's superpoint count, labelled mask, cloud and room ids.
        Global superpoint id = rank * Smax + local id (the all-gather of exchange 2 is padded to Smax rows per rank)."""
        if getattr(self, "_dist", None) is not None and self._dist["comm"] is comm:
            return self._dist
        W = comm.world
        # the labelled rows are drawn from the pool of ALL ranks at once (sampler2.py:294-302 draws over every cloud's labelled regions): exchange the
        # pools' ground-truth dominant labels, order them as one process over the union would meet them (room id, superpoint), draw, keep one's own
        npool = comm.allgather_host(np.array([len(self.lab_pool)], np.int64)).reshape(-1)
        Pmax = max(int(npool.max()), 1)
        def padp(a):
            out = np.full(Pmax, -1, np.int64); out[: len(a)] = a; return out
        pc = self.sp_cloud_h[self.lab_pool]
        g_room = comm.allgather_host(padp(np.asarray(self.room_ids, np.int64)[pc])).reshape(-1)
        g_sp = comm.allgather_host(padp(self.lab_pool - np.asarray(self.sp_base, np.int64)[pc])).reshape(-1)
        g_dom = comm.allgather_host(padp(self.lab_pool_dom)).reshape(-1)
        g_rank = np.repeat(np.arange(W), Pmax); g_loc = np.tile(np.arange(Pmax), W)
        live = np.flatnonzero(g_room >= 0)
        live = live[np.lexsort((g_sp[live], g_room[live]))]
        pos_of = np.full(Pmax, -1, np.int64)
        me = g_rank[live] == comm.rank
        pos_of[g_loc[live][me]] = np.flatnonzero(me)
        self._draw_labelled(self.lab_pool, g_dom[live], pos_of[: len(self.lab_pool)])
        S_all = comm.allgather_host(np.array([self.S, self.B], np.int64))
        Smax = int(S_all[:, 0].max())
        def padded(a, fill):
            out = np.full(Smax, fill, np.int64); out[: self.S] = a; return out
        lab = comm.allgather_host(padded(self.skip_mask.astype(np.int64), 1)).reshape(-1)          # not in the ranked population (labelled / below min_size); padding counts as such
        cloud = comm.allgather_host(padded(self.sp_cloud_h, -1))                                      # local cloud index
        room = comm.allgather_host(padded(np.asarray(self.room_ids, np.int64)[self.sp_cloud_h], -1)).reshape(-1)
        spin = comm.allgather_host(padded(np.arange(self.S) - np.asarray(self.sp_base, np.int64)[self.sp_cloud_h], -1)).reshape(-1)
        Bmax = int(S_all[:, 1].max())
        gcloud = (cloud + (np.arange(W)[:, None] * Bmax)).reshape(-1)                                 # global cloud index, rank-major
        gcloud[cloud.reshape(-1) < 0] = -1
        # sampling()'s batch_size is ONE number for the whole round (ssdr_main_S3DIS2.py:134: 10 000 regions whatever the number of clouds): with
        # batch_size set it stays that, however many ranks share the clouds; the default grows it with the tiles (select_per_tile each)
        batch = self.batch_size if self.batch_size is not None else self.select_per_tile * int(S_all[:, 1].sum())
        # the device-side rule of the sharded run (ssdr_gcn_fps_sharded_local_dev): global labelled mask (padding = labelled), where every global cloud's
        # regions start, and the capacities the static tables bound
        base_l = np.concatenate([np.asarray(self.sp_base, np.int64), np.full(Bmax + 1 - self.B, self.S, np.int64)])      # local, padded to Bmax + 1 entries
        gbase = (comm.allgather_host(base_l[:Bmax]) + np.arange(W)[:, None] * Smax).reshape(-1)
        gbase = np.concatenate([gbase, [W * Smax]]).astype(np.int32)
        nvalid_c = np.array([int((~self.skip_mask[base_l[b]:base_l[b + 1]]).sum()) for b in range(self.B)], np.int64)
        nvalid_all = comm.allgather_host(np.array([int(nvalid_c.sum())], np.int64)).reshape(-1)
        picks = int(min(batch, nvalid_all.sum()))
        T = self._sel_static
        nlab_c = np.diff(T["d_lab_off"].to_host()).astype(np.int64)
        cap_unl = int(min(2 * picks, nvalid_c.sum()))
        share = np.minimum(2 * picks, nvalid_c)
        left, sq = cap_unl, 0
        for i in np.argsort(-(share + nlab_c), kind="stable"):
            a = int(min(share[i], left)); left -= a
            sq += (a + int(nlab_c[i])) ** 2
        nu_dev = max(int(np.minimum(2 * picks, nvalid_all).max()), 1)
        rep = max(int(os.environ.get("SSDR_EMULATE_WORLD", "1")), 1)       # development: the FPS load of `rep` x the ranks (tools/gpu_emulate_world.sh)
        cap_fps = max(rep * int(min(2 * picks, nvalid_all.sum())), 1)
        nlab = comm.allgather_host(np.array([int(T["n_lab"])], np.int64)).reshape(-1)
        # the k-center selector also sends every rank's labelled regions' rows (behind its candidates, padded to nl_max) and seeds the global chain with them
        kc = self.selector == "kcenter"
        nl_max = int(nlab.max()) if kc else 0
        n_lab_all = int(nlab.sum()) if kc else 0
        per = nu_dev + nl_max
        dev = dict(picks=picks, cap_rows=max(cap_unl + int(nlab_c.sum()), 1), cap_nmax=max(int((share + nlab_c).max()), 1), cap_sq=max(int(sq), 1), nu_max=nu_dev,
                   cap_fps=cap_fps, rep=rep, d_glab=DevArray.from_host((lab != 0).astype(np.uint8)), d_gbase=DevArray.from_host(gbase),
                   nl_max=nl_max, n_lab_all=n_lab_all, per=per,
                   d_send=DevArray((per, 32), np.float64), d_gath=DevArray((W, per, 32), np.float64), d_glob=DevArray((cap_fps + n_lab_all + 1, 32), np.float64),
                   d_nlab_off=DevArray.from_host(np.concatenate([[0], np.cumsum(nlab)]).astype(np.int32)), d_already=DevArray((max(n_lab_all, 1),), np.int32),
                   d_plan=DevArray((16 + W + 2 * W * nu_dev,), np.int32), d_out=DevArray((max(rep * picks, 1),), np.int32))
        self._dist = dict(dev=dev, comm=comm, Smax=Smax, Bmax=Bmax, valid=lab == 0, gcloud=gcloud, room=room, spin=spin, batch=batch,
                          S_total=int(S_all[:, 0].sum()), n_pop=int(nvalid_all.sum()), nu_max=int(min(2 * batch, Smax)),
                          nlab=nlab,
                          d_lab=DevArray.from_host(self.skip_mask.astype(np.uint8)),
                          d_masked=DevArray((Smax,), np.float64), d_all=DevArray((W * Smax,), np.float64), d_ord=DevArray((W * Smax,), np.int32))
        return self._dist

    def _score_async(self, comm=None):
        """The scoring stage, enqueued, never waits.  With a communicator the two exchanges run on the same stream, on device buffers:
        all-reduce of the class histogram, all-gather of the (masked, padded) region uncertainties, then the global ranking."""
        cfg, L = self.cfg, _lib.lib()
        n = self.n_pts
        um = {"lc": 0, "entropy": 1, "sb": 2}[[a for a in self.sampler_args if a in ("lc", "entropy", "sb")][0]]
        rm = {"mean": 0, "sum_weight": 1, "WetSU": 2}[[a for a in self.sampler_args if a in ("mean", "sum_weight", "WetSU")][0]]
        st = self.score_stream if self.score_stream is not None else self.stream
        _lib.check(L.ssdr_point_uncertainty_dev(self.probs.ptr, n, cfg.num_classes, um, self.unc.ptr, self.cls.ptr, st))
        _lib.check(L.ssdr_region_stats_dev(self.unc.ptr, self.cls.ptr, self.sp_off.ptr, self.sp_pts.ptr, self.S, cfg.num_classes, rm,
                                           self.region_unc.ptr, self.dom.ptr, self.dom_cnt.ptr, st))
        # the labelled regions' dominant GROUND-TRUTH class (their dominant_point_ids, sampler2.py:288-291) from this batch's tile labels
        _lib.check(L.ssdr_dominant_label_dev(self.tile_l.ptr, self.sp_off.ptr, self.sp_pts.ptr, self.S, max(cfg.num_classes, 1), self.gt_dom.ptr, self.gt_purity.ptr, st))
        # class balance over the ranked population only (prediction() appends only unlabelled regions to region_class, :612-627): "classbal"
        # (add_classbal, :256-260) is the same without the already-selected list and is tested first, as the reference does (:635-638)
        bal = "classbal" in self.sampler_args or "clsbal" in self.sampler_args
        nsel = 0 if "classbal" in self.sampler_args else self.selected_class_list.shape[0]
        T = self._sel_static
        if comm is None:
            if bal:
                _lib.check(L.ssdr_clsbal_dev(self.dom.ptr, self.S, T["d_lab"].ptr, self.selected_class_list.ptr, nsel, self.region_unc.ptr, st))
            _lib.check(L.ssdr_rank_regions_dev(self.region_unc.ptr, self.S, self.sorted_inds.ptr, st))
            self.global_order = None
            return
        D = self._dist_setup(comm)
        if bal:       # exchange 1: the class histogram is global (the already-selected list is counted once, on rank 0)
            _lib.check(L.ssdr_class_hist_dev(self.dom.ptr, self.S, D["d_lab"].ptr, self.selected_class_list.ptr, nsel if comm.rank == 0 else 0, self.hist.ptr, st))
            comm.allreduce_sum_(self.hist, st)
            _lib.check(L.ssdr_clsbal_hist_dev(self.dom.ptr, self.S, self.hist.ptr, D["n_pop"] + nsel, self.region_unc.ptr, st))
        # exchange 2: rank the regions of ALL ranks; labelled regions (and the padding) are -inf and sort last
        _lib.check(L.ssdr_mask_regions_dev(self.region_unc.ptr, D["d_lab"].ptr, self.S, D["Smax"], D["d_masked"].ptr, st))
        comm.allgather_(D["d_masked"], D["d_all"], st)
        _lib.check(L.ssdr_rank_regions_dev(D["d_all"].ptr, comm.world * D["Smax"], D["d_ord"].ptr, st))
        self.global_order = D

    def _score_finish(self, comm=None):
        """kept for callers of the two-step form: the exchanges are enqueued by _score_async and nothing waits on the host any more"""
        return

    def _candidates(self, order, valid, cloud, batch_size):
        """create_file_top_and_all + the candidate rule of sampling() (sampler2.py:533-552, :745-753) on index lists: `order` ranks
        the regions by descending uncertainty, valid[i] = region i may compete (not labelled), cloud[i] = its cloud.  Returns the
        candidates (cloud ascending, descending uncertainty inside a cloud) and their clouds, and the number to select.
        The order is canonical: the reference's own depends on a shuffled DataLoader (sampler2.py:323) and carries no meaning."""
        order = np.asarray(order, np.int64)
        cand = order[valid[order]]                            # labelled regions never compete
        c = cloud[cand]
        nc = int(cloud.max()) + 1 if len(cloud) else 0
        ntop = np.bincount(c[: min(batch_size, len(order))], minlength=nc)      # selected_num per cloud (len(file_list_top[cloud])): the first batch_size candidates are "top"
        # cloud ascending, descending uncertainty inside a cloud (a stable sort of 16-bit keys is a radix sort in NumPy: the sharded run ranks
        # the regions of ALL ranks here, 8 x as many on 8 GPUs)
        grp = np.argsort(c.astype(np.uint16) if nc <= 65536 else c, kind="stable")
        cand, c = cand[grp], c[grp]
        counts = np.bincount(c, minlength=nc)
        first = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64) if nc else np.zeros(0, np.int64)
        pos = np.arange(len(cand)) - first[c]
        take = pos < 2 * ntop[c]                              # candidates = first 2 x selected_num of the cloud (:748)
        return cand[take], c[take], int(ntop.sum())

    def _select(self, comm=None):
        self._select_issue(comm)
        return self._select_collect()

    def _select_issue(self, comm=None):
        """everything of the selection up to the enqueued FPS chain (the host decisions and uploads happen here)"""
        L = _lib.lib()
        st = self.sel_stream          # None: the library stream; a stream of its own lets the selections of consecutive batches overlap
        _lib.check(L.ssdr_select_set_chamfer_mode(self.chamfer_mode))
        T = self._sel_static
        kc = self.selector == "kcenter"
        if (self.global_order is None and T["picks"] > 0 and (not kc or T["n_lab"] > 0)
                and not os.environ.get("SSDR_SELECT_HOST_RULE")):
            # candidate rule + GCN_FPS_sampling enqueued as one chain: the host decides nothing and uploads nothing (the result is read in _select_collect)
            _lib.check(L.ssdr_gcn_fps_sampling_dev(self.f32.ptr, 32, self.cls.ptr, self.dom.ptr, self.tile_l.ptr, self.gt_dom.ptr, self.xyz.ptr, self.sp_off.ptr, self.sp_pts.ptr,
                                                   self.sorted_inds.ptr, self.S, T["d_lab"].ptr, T["d_base"].ptr, self.B, T["d_lab_off"].ptr, T["d_lab_sp"].ptr,
                                                   T["n_lab"], T["batch"], int(self.gcn_number), int(self.gcn_top), 1 if kc else 0, int(self.fps_start), T["cap_rows"], T["cap_nmax"], T["cap_sq"],
                                                   T["cap_unl"], T["picks"], T["d_result"].ptr, st))
            self._pending = ("device", None)
            self.rule_path = "device"
            return
        if (self.global_order is not None and self.global_order["dev"]["picks"] > 0 and (not kc or self.global_order["dev"]["n_lab_all"] > 0)
                and not os.environ.get("SSDR_SELECT_HOST_RULE")):
            # the sharded run, still without a host decision: the rule over the global ranking + this rank's graph, the all-gather of the candidates'
            # propagated features (exchange 3), the replicated global FPS — three enqueues, nothing read back here
            D = self.global_order; V = D["dev"]
            _lib.check(L.ssdr_gcn_fps_sharded_local_dev(self.f32.ptr, 32, self.cls.ptr, self.dom.ptr, self.tile_l.ptr, self.gt_dom.ptr, self.xyz.ptr, self.sp_off.ptr, self.sp_pts.ptr,
                                                        T["d_lab_off"].ptr, T["d_lab_sp"].ptr, T["n_lab"], self.B, D["d_ord"].ptr, comm.world * D["Smax"],
                                                        V["d_glab"].ptr, V["d_gbase"].ptr, comm.rank, comm.world, D["Smax"], D["Bmax"], D["batch"],
                                                        int(self.gcn_number), int(self.gcn_top), V["cap_rows"], V["cap_nmax"], V["cap_sq"], V["nu_max"], V["nl_max"],
                                                        V["d_send"].ptr, V["d_plan"].ptr, st))
            comm.allgather_(V["d_send"], V["d_gath"], st)
            if kc:       # BASELINE configuration 4: [candidates | labelled regions] of all ranks, the labelled ones already selected
                _lib.check(L.ssdr_kcenter_gathered_dev(V["d_gath"].ptr, V["d_plan"].ptr, comm.world, V["nu_max"], V["nl_max"], V["d_nlab_off"].ptr, V["n_lab_all"],
                                                       V["cap_fps"] + V["n_lab_all"] + 1, V["picks"], V["d_glob"].ptr, V["d_already"].ptr, V["d_out"].ptr, st))
            else:
                _lib.check(L.ssdr_fps_gathered_dev(V["d_gath"].ptr, V["d_plan"].ptr, comm.world, V["nu_max"], V["cap_fps"], V["rep"], int(self.fps_start), V["rep"] * V["picks"],
                                                   V["d_glob"].ptr, V["d_out"].ptr, st))
            self._pending = ("sharded", comm)
            self.rule_path = "sharded-device"
            return
        self.rule_path = "host"
        if self.global_order is None:          # (the D2H below runs on the selection stream, which already waits for the scoring stream's work)
            cand, ccloud, sampling_batch = self._candidates(self.sorted_inds.to_host(st), ~self.skip_mask, self.sp_cloud_h, self._sel_static["batch"])
            unl_c, unl_s = np.asarray(ccloud, np.int64), np.asarray(cand, np.int64)
            gl_room = np.asarray(self.room_ids, np.int64)[ccloud]; gl_sp = cand - np.asarray(self.sp_base, np.int64)[ccloud]
            counts_r = None
        else:       # every rank derives the global candidate list from the global ranking, then keeps its own rows
            D = self.global_order
            gcand, gcloud, sampling_batch = self._candidates(D["d_ord"].to_host(st), D["valid"], D["gcloud"], D["batch"])
            r_of = gcand // D["Smax"]
            counts_r = np.bincount(r_of, minlength=comm.world)
            mine = r_of == comm.rank
            unl_c, unl_s = (gcloud[mine] - comm.rank * D["Bmax"]).astype(np.int64), (gcand[mine] - comm.rank * D["Smax"]).astype(np.int64)
            gl_room, gl_sp = D["room"][gcand], D["spin"][gcand]
        unl = list(zip(unl_c.tolist(), unl_s.tolist()))          # (cloud, superpoint) of this rank's candidates
        lab_c, lab_s = T["lab_cloud"], T["lab_sp_h"]              # the labelled regions, cloud by cloud, ascending superpoint id (static)
        sel = np.concatenate([unl_s, lab_s]).astype(np.int32)
        # every cloud's chamfer graph and propagation hop in one batched call (rows grouped cloud by cloud)
        ref_cloud = np.concatenate([unl_c, lab_c])
        order = np.argsort(ref_cloud, kind="stable").astype(np.int32)
        clouds, counts = np.unique(ref_cloud, return_counts=True)
        counts = counts.astype(np.int64)
        coff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        boff = np.concatenate([[0], np.cumsum(counts * counts)]).astype(np.int64)
        ntot, nmax, nsq = int(coff[-1]), int(counts.max()), int(boff[-1])
        n_unl = len(unl_s)
        src = np.zeros(0, np.int32)
        if counts_r is not None:     # exchange 3 is padded to nu_max rows per rank: positions of the real rows in the gathered array
            src = np.concatenate([r * self.global_order["nu_max"] + np.arange(c) for r, c in enumerate(counts_r)]).astype(np.int32)
        # ONE upload for all the small index tables (each separate copy is a host round trip behind the kernels in flight)
        parts = [sel, sel[order], order, coff, boff.view(np.int32), src]
        offs = np.cumsum([0] + [(len(p) + 3) // 4 * 4 for p in parts])           # 16-byte aligned pieces
        pack = np.zeros(offs[-1], np.int32)
        for p, o in zip(parts, offs):
            pack[o:o + len(p)] = p
        d_pack = DevArray.from_host(pack, st)
        d_sel, d_gsel, d_rows, d_coff, d_boff, d_src = (d_pack.ptr + 4 * int(o) for o in offs[:-1])
        rows = max(len(sel), self.global_order["nu_max"] if counts_r is not None else 0)
        d_mf = DevArray((len(sel), 32), np.float32)
        d_v = DevArray((len(sel), 32), np.float64); d_comb = DevArray((rows, 32), np.float64)
        d_tmp = [DevArray(d_v.shape, np.float64), DevArray(d_v.shape, np.float64)]
        # compute_features (sampler2.py:333, :339): the candidates' members by the predicted classes, the labelled regions' by the ground truth (:288-291)
        nu_rows = len(unl_s)
        if nu_rows:
            _lib.check(L.ssdr_segment_mean_features_dev(self.f32.ptr, 32, self.cls.ptr, self.dom.ptr, self.sp_off.ptr, self.sp_pts.ptr, d_sel, nu_rows, d_mf.ptr, st))
        if len(sel) > nu_rows:
            _lib.check(L.ssdr_segment_mean_features_dev(self.f32.ptr, 32, self.tile_l.ptr, self.gt_dom.ptr, self.sp_off.ptr, self.sp_pts.ptr, d_sel + 4 * nu_rows,
                                                        len(sel) - nu_rows, d_mf.ptr + 4 * 32 * nu_rows, st))
        # float32 -> float64 as np.concatenate / np.matmul promote it (V and the running sum comb start as the same values)
        _lib.check(L.ssdr_widen_f32_f64_dev(d_mf.ptr, len(sel) * 32, d_v.ptr, d_comb.ptr, st))
        d_cen = DevArray((ntot, 3), np.float64); d_dir = DevArray((nsq,), np.float64); d_adj = DevArray((nsq,), np.float64)
        _lib.check(L.ssdr_cloud_graph_batch_dev(self.xyz.ptr, self.sp_off.ptr, self.sp_pts.ptr, d_gsel, d_coff, d_boff, len(clouds), ntot, nmax,
                                                int(self.gcn_top), d_cen.ptr, d_dir.ptr, d_adj.ptr, st))
        src_v = d_v
        for hop in range(int(self.gcn_number)):
            dst = d_tmp[hop & 1]
            _lib.check(L.ssdr_propagate_batch_dev(d_adj.ptr, d_coff, d_boff, len(clouds), nmax, d_rows, src_v.ptr, 32, dst.ptr, d_comb.ptr, st))
            src_v = dst
        keep = [d_pack, d_cen, d_dir, d_adj, d_v, d_tmp, d_mf, d_comb]
        self.unl_cloud_ids, self.unl_sp = gl_room, gl_sp          # (room id, superpoint inside its room) of every candidate, global order
        n_lab = len(lab_s)
        if comm is not None:                                 # exchange 3: the candidates' propagated features, on the selection stream
            D = self.global_order
            kc = self.selector == "kcenter"                  # k-center also needs the labelled regions' rows of every rank
            nl_max = int(D["nlab"].max()) if kc else 0
            per = D["nu_max"] + nl_max
            d_send = d_comb
            if kc:                                           # [candidates, padded to nu_max | labelled, padded to nl_max]
                send_idx = np.zeros(per, np.int32)
                send_idx[: len(unl)] = np.arange(len(unl)); send_idx[D["nu_max"]: D["nu_max"] + n_lab] = len(unl) + np.arange(n_lab)
                d_sidx = DevArray.from_host(send_idx, st); d_send = DevArray((per, 32), np.float64)
                _lib.check(L.ssdr_gather_rows_dev(d_comb.ptr, d_sidx.ptr, per, 32 * 8, d_send.ptr, st))
                keep += [d_sidx, d_send]
                src = np.concatenate([r * per + np.arange(c) for r, c in enumerate(counts_r)] +
                                     [r * per + D["nu_max"] + np.arange(c) for r, c in enumerate(D["nlab"])]).astype(np.int32)
                d_src_a = DevArray.from_host(src, st); d_src = d_src_a.ptr; keep.append(d_src_a)
                n_lab = int(D["nlab"].sum())
            d_gath = DevArray((comm.world, per, 32), np.float64)
            comm.allgather_(_Prefix(d_send, per * 32), d_gath, st)
            n_unl = int(counts_r.sum())
            n_rows = n_unl + (n_lab if kc else 0)
            d_glob = DevArray((max(n_rows, 1), 32), np.float64)
            _lib.check(L.ssdr_gather_rows_dev(d_gath.ptr, d_src, n_rows, 32 * 8, d_glob.ptr, st))
            keep += [d_gath, d_glob]
            d_comb = d_glob
            self._comb_dev, self._comb_n = d_glob, n_rows
        emu = int(os.environ.get("SSDR_EMULATE_WORLD", "0"))
        if emu > 1 and comm is not None and self.selector != "kcenter":
            # development: the FPS LOAD of `emu` ranks on one GPU (the replicated global chain is N^2: N x the rows, N x the picks) — the rows are
            # repeated, the picks beyond the real ones are ties; only the timing means anything (tools/gpu_emulate_world.sh)
            idx = np.tile(np.arange(n_unl, dtype=np.int32), emu)
            d_idx = DevArray.from_host(idx, st); d_big = DevArray((emu * n_unl, 32), np.float64)
            _lib.check(L.ssdr_gather_rows_dev(d_comb.ptr, d_idx.ptr, emu * n_unl, 32 * 8, d_big.ptr, st))
            keep += [d_idx, d_big]
            d_comb, n_unl, sampling_batch = d_big, emu * n_unl, emu * sampling_batch
            self._emu_mod = len(idx) // emu
        d_out = DevArray((sampling_batch,), np.int32)
        if self.selector == "kcenter":
            d_already = DevArray.from_host((n_unl + np.arange(n_lab)).astype(np.int32), st); keep.append(d_already)
            _lib.check(L.ssdr_kcenter_dev(d_comb.ptr, n_unl + n_lab, 32, d_already.ptr, n_lab, sampling_batch, d_out.ptr, st))
        else:
            _lib.check(L.ssdr_fps_dev(d_comb.ptr, n_unl, 32, int(self.fps_start), sampling_batch, d_out.ptr, st))
        self._keep = keep
        self._pending = (d_out, unl)

    @property
    def comb_all(self):
        """the gathered candidate features of the last sharded step (tests)"""
        _lib.sync()
        return self._comb_dev.to_host()[: self._comb_n]

    def _select_collect(self):
        """wait for the FPS chain of _select_issue and read the selection back"""
        d_out, unl = self._pending
        self._pending = None
        if isinstance(d_out, str) and d_out == "sharded":     # the sharded device-side rule: the plan (counts, global candidate list) and the picks
            comm = unl
            D = self.global_order; V = D["dev"]; W = comm.world
            plan = V["d_plan"].to_host(self.sel_stream)          # waits for the selection stream alone
            sel = V["d_out"].to_host(self.sel_stream)
            _lib.check(_lib.lib().ssdr_select_status(self.sel_stream, None))
            if plan[5] or plan[9]:
                raise RuntimeError("gcn_fps_sharded: the candidate rule produced more rows than the capacities allow (status %d, %d)" % (int(plan[5]), int(plan[9])))
            n_g = int(plan[8])
            gcand = plan[16 + W + W * V["nu_max"]: 16 + W + W * V["nu_max"] + n_g].astype(np.int64)
            mine = gcand // D["Smax"] == comm.rank
            loc = gcand[mine] - comm.rank * D["Smax"]
            unl = list(zip(self.sp_cloud_h[loc].tolist(), loc.tolist()))
            self.unl_cloud_ids, self.unl_sp = D["room"][gcand], D["spin"][gcand]
            self._comb_dev, self._comb_n = V["d_glob"], n_g + (V["n_lab_all"] if self.selector == "kcenter" else 0)
            if V["rep"] > 1:
                self._emu_mod = n_g
        elif isinstance(d_out, str):                          # the device-side rule: counts, picks and the candidate list in one read-back
            T = self._sel_static
            res = T["d_result"].to_host(self.sel_stream)     # waits for the selection stream alone
            # from ~20 tiles per GPU on the chain's FPS / k-center is a cooperative launch: one that was not co-resident reports it here
            _lib.check(_lib.lib().ssdr_select_status(self.sel_stream, None))
            if res[5]:
                raise RuntimeError("gcn_fps_sampling: the candidate rule produced more rows than the capacities allow (status %d)" % int(res[5]))
            n_unl, picks = int(res[0]), int(res[4])
            sel = res[8:8 + picks].copy()
            cand = res[8 + T["picks"]: 8 + T["picks"] + n_unl].astype(np.int64)
            ccloud = self.sp_cloud_h[cand]
            unl = _Pairs(ccloud, cand)                               # (20 000 candidates in one AL round: the tuples are built when somebody reads them)
            self.unl_cloud_ids = np.asarray(self.room_ids, np.int64)[ccloud]; self.unl_sp = cand - np.asarray(self.sp_base, np.int64)[ccloud]
        else:
            sel = d_out.to_host(self.sel_stream)             # waits for the selection stream alone
            # a cooperative (multi-workgroup) FPS / k-center launch that was not co-resident reports it here instead of returning a wrong selection
            _lib.check(_lib.lib().ssdr_select_status(self.sel_stream, None))
        # the device-flavour KNN calls cannot report what their kernels found (overflowed kd queue / node table / level limit, hand-over
        # list): ask once per batch, here where the host waits anyway — without waiting for the pyramids of the LATER batches that the
        # KNN stream already holds (the finished calls' tickets are looked at; Pipelined.finish / the sequential step wait for all)
        from . import knn as _knn
        _knn.knn_status(self.knn_stream if self.knn_stream is not None else self.stream, wait=not self.pipelined)
        if not self.pipelined:                               # sequential use: the front end of this batch has finished, ask it too
            _lib.check(_lib.lib().ssdr_grid_subsample_status(self.front_stream if self.front_stream is not None else self.stream, None))
        if len(sel) and int(np.min(sel)) < 0:                # (an aborted chain leaves -1 picks: never index the candidate list with them)
            raise RuntimeError("selection: the FPS / k-center chain left unset picks (an aborted cooperative launch)")
        if getattr(self, "_emu_mod", 0):                    # (SSDR_EMULATE_WORLD: the picks index the repeated rows)
            sel = sel % self._emu_mod
        si = np.asarray(sel, np.int64)
        self._selected = _Pairs(np.asarray(self.unl_cloud_ids)[si], np.asarray(self.unl_sp)[si])      # (room id, superpoint in room)
        return sel, unl

    @property
    def selected(self):
        """the picks as [(room id, superpoint in room), ...] — a plain list, built on first access"""
        v = self.__dict__.get("_selected")
        return v.tolist() if isinstance(v, _Pairs) else v

    @selected.setter
    def selected(self, value):
        self._selected = value

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


class _Rows:
    """rows [lo, hi) of a DevArray's first axis with the surface the stages use (ptr / shape / dtype / to_host): the batches of an AL round write
    their tiles and network outputs straight into their slice of the round's arrays"""
    def __init__(self, arr, lo, hi):
        rowb = arr.dtype.itemsize * int(np.prod(arr.shape[1:], dtype=np.int64))
        self.base, self.dtype, self.shape = arr, arr.dtype, (int(hi - lo),) + tuple(arr.shape[1:])
        self.ptr, self.nbytes = int(arr.ptr) + int(lo) * rowb, int(hi - lo) * rowb

    def to_host(self, stream=None):
        out = np.empty(self.shape, self.dtype)
        if stream is None:
            _lib.check(_lib.lib().ssdr_memcpy_d2h(_lib.ptr(out), self.ptr, self.nbytes))
        else:
            _lib.check(_lib.lib().ssdr_memcpy_d2h_on(_lib.ptr(out), self.ptr, self.nbytes, stream))
        return out


class ALRound:
    """ONE active-learning round at the reference's own scale: the network runs over ALL clouds of the pool and then ONE selection picks
    `batch_size` regions among 2 x batch_size candidates (+ the labelled rows) of all of them (ssdr_main_S3DIS2.py:134: 10 000 regions per round;
    sampler2.py:580-642 prediction over every cloud, :736-781 one GCN_FPS_sampling).  bench.py's step selects per 16-tile batch instead, which
    keeps the reference's picks-per-tile ratio but under-represents the quadratic term of the farthest-point chain 17-fold; this class is the
    round as the reference runs it: `n_batches` batches of len(rooms) tiles go through front end -> KNN pyramid -> inference (three streams, three
    buffer sets, batches overlapped), their tiles / labels / probabilities / features land in the round's arrays, then scoring over all points and
    the one-call device chain (ssdr_gcn_fps_sampling_dev) over all clouds' regions.  Tiles of batch b are cut from the same raw rooms with the
    randomness of room id b * len(rooms) + i (another pick point, shuffle and padding draw: another tile)."""
    SLOTS = 4          # batches in flight = HIP streams = the runtime's hardware queues (2: 54.9 ms for 17 batches, 3: 52.2, 4: 49.1, 5: 56.9, 6: 49.4, 8: 49.6; a stream per STAGE: 60-64)

    def __init__(self, weights, rooms, n_batches, config=ConfigS3DIS, batch_size=10000, round_num=5, labeled_per_tile=15, precision="f32",
                 selector="fps", tiles32=True, seed=0, gcn_number=1, gcn_top=0, min_size=1):
        L = _lib.lib()
        self.cfg, self.nb, self.B, self.rooms = config, int(n_batches), len(rooms), rooms
        N = config.num_points
        self.tiles = self.nb * self.B
        # five streams: front end and pyramid alternate between two streams each by the parity of the batch, so that a cross-stream wait ("everything
        # the producer holds so far", ssdr_stream_wait) names exactly the batch it is meant for — with one stream per stage one of the three waits of a
        # step always caught the neighbouring batch as well, and the stages overlapped two deep instead of three
        # One stream per stage.  A consumer stage waits for "everything its producer stream holds so far" (ssdr_stream_wait), which names exactly its own
        # batch because a step enqueues consumer first (inference k - 2, pyramid k - 1, front end k); the front end's wait for the LAST READER of the buffer
        # set it reuses (inference k - SLOTS) is an event recorded behind that inference (ssdr_event_*): with three buffer sets and a stream-wide wait the
        # front end ran at most two batches ahead and the 17 batches took 60 ms; five sets and the event: the stages run as far ahead as their inputs allow.
        self.streams = []
        for _ in range(3):
            st = C.c_void_p(); _lib.check(L.ssdr_stream_create(C.byref(st))); self.streams.append(st.value)
        self.s_front, self.s_knn, self.s_inf = self.streams
        self.SLOTS = int(os.environ.get("SSDR_AL_SLOTS", self.SLOTS))
        self.bstreams = []
        for _ in range(self.SLOTS):
            st = C.c_void_p(); _lib.check(L.ssdr_stream_create(C.byref(st))); self.bstreams.append(st.value)
        self.ev_inf = []
        for _ in range(self.nb):
            e = C.c_void_p(); _lib.check(L.ssdr_event_create(C.byref(e))); self.ev_inf.append(e.value)
        s_i = self.s_inf
        self.work = []
        for w in range(self.SLOTS):
            h = HotPath(weights, config, precision=precision, tiles32=tiles32, seed=seed, select_per_tile=1, labeled_per_tile=1)
            h.front_stream, h.knn_stream, h.stream, h.pipelined = self.s_front, self.s_knn, s_i, True
            h.load_rooms(rooms, list(range(self.B)))
            self.work.append(h)
        P = self.tiles * N
        self.xyz = DevArray((P, 3), np.float32); self.tile_l = DevArray((P,), np.int32)
        self.probs = DevArray((P, config.num_classes), np.float32); self.f32 = DevArray((P, 32), np.float32)
        # per batch: the tiles' randomness and where its outputs go
        self.batches = []
        h0 = self.work[0]
        for b in range(self.nb):
            draws = [h0.draw_room(r[0], b * self.B + i) for i, r in enumerate(rooms)]
            lo, hi = b * self.B * N, (b + 1) * self.B * N
            self.batches.append(dict(centers=np.ascontiguousarray(np.stack([d["center"] for d in draws]), np.float32),
                                     perm=DevArray.from_host(np.stack([d["perm"] for d in draws])), dup=DevArray.from_host(np.stack([d["dup"] for d in draws])),
                                     xyz=_Rows(self.xyz, lo, hi), tile_l=_Rows(self.tile_l, lo, hi), probs=_Rows(self.probs, lo, hi), f32=_Rows(self.f32, lo, hi)))
        # setup (untimed): every batch's tiles once, their superpoints (stand-in for the partition, as HotPath.load_rooms), the labelled stand-in
        for b in range(self.nb):
            self._bind(b)._front_end()
        _lib.sync(self.s_front)
        _lib.check(L.ssdr_grid_subsample_status(self.s_front, None))
        from .synthetic import superpoints_from_tile
        tiles = self.xyz.to_host().reshape(self.tiles, N, 3)
        offs, pts, cloud, labeled = [np.zeros(1, np.int64)], [], [], {}
        for t in range(self.tiles):
            o, p = superpoints_from_tile(tiles[t])
            base = int(sum(len(x) for x in cloud))
            pts.append(p.astype(np.int64) + t * N); offs.append(o[1:].astype(np.int64) + offs[-1][-1]); cloud.append(np.full(len(o) - 1, t, np.int32))
            rng = np.random.default_rng([seed, t, 1])
            n_sp = len(o) - 1
            labeled[t] = set((base + rng.choice(n_sp, min(labeled_per_tile, n_sp), replace=False)).tolist())
        sel_list = np.random.default_rng([seed, 999983]).integers(0, config.num_classes, 4000)
        self.sel = HotPath.from_device(self.xyz, self.probs, self.f32, self.tile_l, np.concatenate(offs), np.concatenate(pts), np.concatenate(cloud), labeled, sel_list,
                                       config, batch_size=batch_size, round_num=round_num, selector=selector, gcn_number=gcn_number, gcn_top=gcn_top, min_size=min_size, seed=seed)
        self.sel.stream = self.sel.score_stream = self.sel.sel_stream = s_i
        self.sel.front_stream = s_i; self.sel.pipelined = True
        self.tile_points = P

    def _bind(self, b):
        """worker of batch b with the batch's randomness and output slices"""
        h, d = self.work[b % self.SLOTS], self.batches[b]
        h.centers, h.perm, h.dup = d["centers"], d["perm"], d["dup"]
        h.xyz, h.tile_l, h.probs, h.f32 = d["xyz"], d["tile_l"], d["probs"], d["f32"]
        return h

    def infer_all(self):
        """front end -> KNN pyramid -> inference of every batch, enqueued.  A BATCH PER STREAM (default): batch b runs its three stages in order on stream
        b mod SLOTS, whose previous batch used the same buffer set — no cross-stream wait at all, and what overlaps is whatever the SLOTS batches in flight
        have to offer each other (a pyramid's latency-bound tree hand-over beside the next batch's grid search as well as beside another stage).
        SSDR_AL_SCHED=stage: a stream per stage with producer waits and an event for the buffer set's last reader (measured slower, tools/al_probe.py)."""
        L = _lib.lib()
        if os.environ.get("SSDR_AL_SCHED", "batch") == "batch":
            for b in range(self.nb):
                h = self._bind(b)
                st = self.bstreams[b % self.SLOTS]
                h.front_stream = h.knn_stream = h.stream = st
                if b < self.SLOTS:
                    _lib.check(L.ssdr_stream_wait(st, self.s_inf))      # (the previous round's selection, which read these arrays, has finished)
                h._front_end(); h._pyramid(); h._infer()
            for st in self.bstreams[: min(self.SLOTS, self.nb)]:
                _lib.check(L.ssdr_stream_wait(self.s_inf, st))          # the selection (on the inference stream) starts after every batch
            return
        s_f, s_k, s_i = self.s_front, self.s_knn, self.s_inf
        nowait = bool(os.environ.get("SSDR_AL_NOSLOTWAIT"))       # (development, timing only: a front end may then overwrite buffers an inference still reads)
        _lib.check(L.ssdr_stream_wait(s_f, s_i))                  # (a previous round's inferences, and the selection that read their outputs, have finished)
        for k in range(self.nb + 2):
            if 0 <= k - 2 < self.nb:
                h = self._bind(k - 2); h.stream = s_i
                _lib.check(L.ssdr_stream_wait(s_i, s_k))          # (the pyramid stream's newest work is batch k - 2's)
                h._infer()
                _lib.check(L.ssdr_event_record(self.ev_inf[k - 2], s_i))
            if 0 <= k - 1 < self.nb:
                h = self._bind(k - 1); h.knn_stream = s_k
                _lib.check(L.ssdr_stream_wait(s_k, s_f))          # (the front stream's newest work is batch k - 1's)
                h._pyramid()
            if k < self.nb:
                h = self._bind(k); h.front_stream = s_f
                if k >= self.SLOTS and not nowait:                # the buffer set's last reader: the inference of batch k - SLOTS
                    _lib.check(L.ssdr_stream_wait_event(s_f, self.ev_inf[k - self.SLOTS]))
                h._front_end()

    def run(self):
        """the whole round; returns (picked candidate indices, candidate list) as HotPath.step does"""
        self.infer_all()
        self.sel._score_async(None)
        self.sel._select_issue(None)
        out = self.sel._select_collect()
        from . import knn as _knn
        for st in [self.s_knn] + self.bstreams:
            _knn.knn_status(st)
            _lib.check(_lib.lib().ssdr_grid_subsample_status(st, None))
        _lib.check(_lib.lib().ssdr_grid_subsample_status(self.s_front, None))
        return out


class BatchStreams:
    """`slots` batches in flight, A BATCH PER STREAM: batch k runs front end -> KNN pyramid -> network -> scoring -> selection in order on stream
    k mod slots, on the buffer set that stream's previous batch used.  No cross-stream wait exists; what overlaps is whatever the batches in flight
    have to offer each other (a selection's one-workgroup chain beside another batch's network, a pyramid's tree hand-over beside a front end).
    Against Pipelined (a stream per STAGE) there is no fill — every stream is busy from the first launch on — and the drain is the last batches'
    selections; the host waits for the selection of batch k - slots before it reuses its buffers.  Same interface as Pipelined for bench.py."""

    def __init__(self, make_hot_path, slots=4, sel_streams=0):
        """sel_streams > 0: the selections run on that many streams of their own, taken in turn (a selection waits for its batch's stream; the batch's
        stream goes on to its next batch only after the host has read that selection: the buffer set is the same)"""
        L = _lib.lib()
        self.depth = self.slots = int(slots)

        def mkstream():
            st = C.c_void_p(); _lib.check(L.ssdr_stream_create(C.byref(st))); return st.value
        self.streams = [mkstream() for _ in range(self.slots)]
        self.sel_streams = [mkstream() for _ in range(int(sel_streams))]
        self.hp = [make_hot_path() for _ in range(self.slots)]
        for h, st in zip(self.hp, self.streams):
            h.pipelined = True
            h.front_stream = h.knn_stream = h.stream = h.score_stream = h.sel_stream = st
        self._issued = []
        self._k = 0
        self.finish()

    def run(self, steps, comm=None, steady=False):
        """finishes `steps` selections (every batch issued is completed before the call returns)"""
        out = None
        for k in range(steps):
            h = self.hp[k % self.slots]
            if len(self._issued) >= self.slots:              # the buffer set's previous batch: its result is read before the set is reused
                out = self._issued.pop(0)._select_collect()
            h._front_end(); h._pyramid(); h._infer(); h._score_async(comm)
            if self.sel_streams:
                h.sel_stream = self.sel_streams[self._k % len(self.sel_streams)]
                _lib.check(_lib.lib().ssdr_stream_wait(h.sel_stream, h.stream))
            h._select_issue(comm)
            self._issued.append(h); self._k += 1
        while self._issued:
            out = self._issued.pop(0)._select_collect()
        return out

    def finish(self):
        while self._issued:
            self._issued.pop(0)._select_collect()
        _lib.sync()
        for st in self.streams + self.sel_streams:
            _lib.sync(st)
        from . import knn as _knn
        for st in self.streams:
            _knn.knn_status(st)
            _lib.check(_lib.lib().ssdr_grid_subsample_status(st, None))


class _Prefix:
    """the first `count` elements of a DevArray as a flat array of its own (exchange buffers)"""
    def __init__(self, arr, count):
        self.base = arr          # (the buffer stays the array's: a prefix that outlived it would point into the pool of freed buffers)
        self.ptr, self.dtype, self.shape, self.nbytes = arr.ptr, arr.dtype, (int(count),), int(count) * arr.dtype.itemsize
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.ptr), False), "version": 2, "strides": None}

    def host_view(self):
        import ctypes as C
        return np.frombuffer((C.c_char * self.nbytes).from_address(int(self.ptr)), self.dtype)


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

    def __init__(self, make_hot_path, depth=5, groups=None, overlap_select=True, sel_lag=1, spare_set=False):
        """groups: optional stage -> stream-group tuple for (front, knn, infer, score), non-decreasing from 0; depth = last group + 2"""
        if groups is not None:
            depth = groups[-1] + 2
        assert groups is not None or depth in self.GROUPS
        L = _lib.lib()
        self.depth = depth
        self.group = dict(zip(self.STAGES, groups if groups is not None else self.GROUPS[depth]))
        # The runtime spreads the streams over its (4) hardware queues in the order they are created, after the NULL stream and the
        # library's own (both brought into use by ssdr_init); streams that share a queue serialise.  Which stage shares with which decides
        # the overlap: measured on MI355X / ROCm 7.2 at depth 5 with one unused stream created in front of the stage streams 4.8 ms per
        # step, without it 6.4, with two 6.1, with three 5.3, with the unused one in any later position 5.7-6.2 (`GPU_MAX_HW_QUEUES=8`, every
        # stream on a queue of its own: 4.8 as well).  So: one spare in front.
        def mkstream():
            st = C.c_void_p()
            _lib.check(L.ssdr_stream_create(C.byref(st)))
            return st.value
        qmap = os.environ.get("SSDR_PIPE_QMAP")              # development: "f,k,i,s,a,b" = the hardware queue (1..GPU_MAX_HW_QUEUES) wanted for the front / knn / infer /
        if qmap and depth == 5 and overlap_select:           # score streams and the two selection streams ("-" for a: the library stream, queue 1)
            # ROCm 7.2's runtime (read off rocprofv3's queue ids, tools/gpu_qdiscover.sh): the library's stream holds queue 1, the NULL stream queue 2, every new stream
            # opens a new queue until GPU_MAX_HW_QUEUES (4) exist and then joins the queue with the fewest streams, the HIGHEST id among equals; unused streams count
            want = qmap.split(",")
            maxq = int(os.environ.get("GPU_MAX_HW_QUEUES") or 4)
            refs = {1: 1, 2: 1}
            self._spares = []

            def next_queue():
                if len(refs) < maxq:
                    return len(refs) + 1
                low = min(refs.values())
                return max(q for q, r in refs.items() if r == low)

            def on_queue(q):
                q = int(q)
                for _ in range(64):
                    nq = next_queue()
                    refs[nq] = refs.get(nq, 0) + 1
                    st = mkstream()
                    if nq == q:
                        return st
                    self._spares.append(st)
                raise RuntimeError("SSDR_PIPE_QMAP: queue %d not reachable" % q)
            order = sorted(range(6), key=lambda i: 0)      # creation in the order given: front, knn, infer, score, selA, selB
            made = {}
            for i in order:
                made[i] = None if want[i] == "-" else on_queue(want[i])
            self.streams = [made[0], made[1], made[2], made[3]]
            self.sel_streams = [made[4], made[5]]
            self._spare = None
        else:
            self._spare = mkstream()
            self.streams = [mkstream() for _ in range(depth - 1)]
            self.sel_streams = [None]                        # selections alternate between the library stream and one of their own
            if overlap_select:
                self.sel_streams.append(mkstream())
        self.overlap_select = overlap_select
        # sel_lag: how many selections behind the newest the host waits for (1: the previous batch's).  The sharded run's replicated global FPS
        # grows with the square of the rank count: from 4 ranks on a chain is longer than two steps, and sel_lag = 2 keeps
        # three chains in flight on three streams, with one more buffer set so that no stage overwrites what a running selection reads
        self.sel_lag = max(1, int(sel_lag)) if overlap_select else 1
        for _ in range(self.sel_lag - 1):
            self.sel_streams.append(mkstream())
        self._uncollected = None
        self._issued = []                                    # selections enqueued, not collected yet (oldest first)
        self.lead = {n: depth - 1 - g for n, g in self.group.items()}      # batches ahead of the selection
        # one buffer set per batch in flight (a spare set, so that no stage has to wait for the previous selection's buffers, was
        # measured slower: 127 vs 137 Mpoints/s — a sixth working set in the caches costs more than the deferred front end)
        self.slots = depth + self.sel_lag - 1 + (1 if spare_set else 0)      # spare_set: nothing is deferred behind the wait for the previous selection
        self.hp = [make_hot_path() for _ in range(self.slots)]
        for h in self.hp:
            h.pipelined = True
            h.front_stream, h.knn_stream = self.streams[self.group["front"]], self.streams[self.group["knn"]]
            h.stream, h.score_stream = self.streams[self.group["infer"]], self.streams[self.group["score"]]
        self._k = 0
        self._drain()

    def _drain(self):
        _lib.sync()
        for st in self.streams + [x for x in self.sel_streams if x is not None]:
            _lib.sync(st)

    def _stage(self, name, b):
        """enqueue one stage of batch b; a consumer stream first waits for everything its producer stream holds so far"""
        h = self.hp[b % self.slots]
        i = self.STAGES.index(name)
        if i > 0 and self.group[name] != self.group[self.STAGES[i - 1]]:
            _lib.check(_lib.lib().ssdr_stream_wait(self.streams[self.group[name]], self.streams[self.group[self.STAGES[i - 1]]]))
        if name == "score":
            h._score_async(self.comm)
        else:
            {"front": h._front_end, "knn": h._pyramid, "infer": h._infer}[name]()

    def run(self, steps, comm=None, steady=False):
        """Finishes `steps` selections.  steady=False: fills the pipe, runs, drains (every issued batch is completed).
        steady=True keeps the pipe full across calls: the first call fills it, and every later step issues exactly one launch sequence
        of EVERY stage (on consecutive batches) and completes one selection — what bench.py times; finish() drains.

        The selection chain (host decisions, ~3 ms of dependent launches ending in the one-workgroup FPS) is the longest stage and the only
        one the host waits for.  With overlap_select the host enqueues the selection of batch k, then the other stages, and only then
        waits for the selection of batch k - 1 (alternating selection streams): the chains of consecutive batches overlap and the step
        is no longer the latency of one chain.  The stage that reuses the buffer set of batch k - 1 is enqueued after that wait."""
        self.comm = comm
        L = _lib.lib()
        out = None
        lead, first = self.lead, self.lead["front"]
        k0 = self._k if steady else 0
        last = None if steady else steps                     # batches >= last are never issued
        if k0 == 0:
            self._uncollected = None; self._issued = []
            for b in range(first if steady else min(steps, first)):      # prologue: fill the pipe
                for name in self.STAGES:
                    if b < lead[name]:
                        self._stage(name, b)
        for k in range(k0, k0 + steps):
            hk = self.hp[k % self.slots]
            hk.sel_stream = self.sel_streams[k % len(self.sel_streams)] if self.overlap_select else None
            _lib.check(L.ssdr_stream_wait(hk.sel_stream, self.streams[self.group["score"]]))    # selection stream: batch k's scores are ready
            hk._select_issue(comm)                           # batch k: host decisions + the whole selection chain enqueued ...
            deferred = []
            for b in range(k + 1, k + first + 1):            # ... the other stages are issued while its FPS chain (one workgroup,
                if last is not None and b >= last:           # ~1.4 ms) runs, instead of before it ...
                    break
                for name in self.STAGES:
                    if b == k + lead[name]:
                        if self.overlap_select and b - k == self.slots - self.sel_lag:      # (never with a spare set: b - k < depth)
                            deferred.append((name, b))       # writes the buffer set of batch k - sel_lag, whose selection may still run
                        else:
                            self._stage(name, b)             # the buffer set of batch b was last read by select(b - depth), done
            if self.overlap_select:
                self._issued.append(k)
                while len(self._issued) > self.sel_lag:
                    out = self.hp[self._issued.pop(0) % self.slots]._select_collect()
                self._uncollected = self._issued[0] if self._issued else None
                for name, b in deferred:
                    self._stage(name, b)
            else:
                out = hk._select_collect()                   # ... and only then the host waits for the selection
        if steady:
            self._k = k0 + steps
            if out is None and self._issued:                       # a short call right after the fill: nothing older to hand back
                while self._issued:
                    out = self.hp[self._issued.pop(0) % self.slots]._select_collect()
                self._uncollected = None
        else:
            while self.overlap_select and self._issued:
                out = self.hp[self._issued.pop(0) % self.slots]._select_collect()
            self._uncollected = None
            self._drain()
        return out

    def finish(self):
        """end a steady run: wait for everything issued and forget the partially processed batches"""
        while getattr(self, "_issued", None):
            self.hp[self._issued.pop(0) % self.slots]._select_collect()
        self._uncollected = None
        self._drain()
        from . import knn as _knn
        _knn.knn_status(self.streams[self.group["knn"]])     # everything has finished: the blocking checks
        _lib.check(_lib.lib().ssdr_grid_subsample_status(self.streams[self.group["front"]], None))
        self._k = 0

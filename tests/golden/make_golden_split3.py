"""Golden vectors for the Semantic3D octant partition, produced by the reference's OWN method
(/root/reference/SSRD_AL_semantic3d/semantic3d_dataset_sampling.py: Semantic3D_Dataset_Sampling.split3 + the merge loop of tf_map,
:243-255, restated here in six lines because it sits inside a 60-line method):   python tests/golden/make_golden_split3.py

The clouds are regenerated from seeds by the test (tests/test_split3.py: cloud()); the vectors hold, per combined part, its size and two
checksums of its index SET (the order inside a part is CPython's set iteration order in the reference, see oracle/split3_np.py) — and
the full index arrays for the small case."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/SSRD_AL_semantic3d"
sys.path.insert(0, os.path.dirname(HERE))


def main():
    for n in ("open3d", "open3d.linux", "torchvision", "torchvision.transforms", "PIL", "PIL.Image", "cpp_wrappers", "cpp_wrappers.cpp_subsampling",
              "cpp_wrappers.cpp_subsampling.grid_subsampling", "nearest_neighbors", "nearest_neighbors.lib",
              "nearest_neighbors.lib.python", "nearest_neighbors.lib.python.nearest_neighbors"):
        sys.modules[n] = types.ModuleType(n)
    sys.modules["torchvision.transforms"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    sys.path.insert(0, REF); sys.path.insert(0, os.path.join(REF, "utils"))
    cwd = os.getcwd(); os.chdir(REF)
    try:
        import semantic3d_dataset_sampling as S
    finally:
        os.chdir(cwd)
    from test_split3 import CASES, cloud
    ds = object.__new__(S.Semantic3D_Dataset_Sampling)
    g = {}
    for name, (seed, n, max_size, merge_max) in CASES.items():
        xyz = cloud(seed, n)
        part_list = []
        ds.split3(xyz, np.arange(n), part_list, max_size=max_size)
        comb = []                                    # tf_map :243-251
        for part in part_list:
            if len(part) > merge_max:
                comb.append(part)
            elif len(comb) > 0:
                comb[-1] = np.concatenate([comb[-1], part], axis=0)
            else:
                comb.append(part)
        comb = [np.asarray(p, np.int64) for p in comb if len(list(p)) > 0]          # :254-255
        g[name + "/raw_sizes"] = np.asarray([len(p) for p in part_list], np.int64)
        g[name + "/sizes"] = np.asarray([len(p) for p in comb], np.int64)
        g[name + "/sum"] = np.asarray([int(p.sum()) for p in comb], np.int64)
        g[name + "/sumsq"] = np.asarray([int((p.astype(np.uint64) ** 2).sum() % (1 << 62)) for p in comb], np.int64)
        if n <= 20000:
            g[name + "/sorted"] = np.concatenate([np.sort(p) for p in comb]).astype(np.int32)
        print(name, "raw parts", g[name + "/raw_sizes"].tolist(), "combined", g[name + "/sizes"].tolist())
    np.savez_compressed(os.path.join(HERE, "split3_golden.npz"), **g)
    print("split3_golden.npz", os.path.getsize(os.path.join(HERE, "split3_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

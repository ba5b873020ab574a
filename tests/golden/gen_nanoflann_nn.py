"""Builds tests/golden/nanoflann_nn.npz: findNNDepth known answers from the reference's own vendored nanoflann.

    python tests/golden/gen_nanoflann_nn.py          (in the build container: needs /root/reference)

Compiles gen_nanoflann_nn.cpp against /root/reference/vins_estimator/lib/nanoflann/nanoflann.hpp where it lies (nothing of the
reference is copied), runs it, and stores per cloud: the points, their depths, the queries, and for every query the index nanoflann
returned, its squared distance and the depth findNNDepth hands back.  The fixture is data only; it travels to the GPU box."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INC = "/root/reference/vins_estimator/lib/nanoflann"


def main():
    if not os.path.exists(os.path.join(INC, "nanoflann.hpp")):
        sys.exit("needs the reference tree (build container only)")
    exe = os.path.join(tempfile.gettempdir(), "gen_nanoflann_nn")
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-I", INC, os.path.join(HERE, "gen_nanoflann_nn.cpp"), "-o", exe])
    for arg, name in (("1", "nanoflann_nn.npz"), ("2", "nanoflann_nn2.npz")):   # 2: the degenerate clouds (see the .cpp)
        write(subprocess.check_output([exe, arg], text=True).split(), name)


def write(tok, name):
    it = iter(tok)
    n_clouds = int(next(it))
    out = {"n_clouds": np.int64(n_clouds)}
    for c in range(n_clouds):
        n, nq = int(next(it)), int(next(it))
        cl = np.array([[float(next(it)) for _ in range(3)] for _ in range(n)]).reshape(n, 3)
        qs = np.array([[float(next(it)) for _ in range(5)] for _ in range(nq)]).reshape(nq, 5)
        out[f"cloud_xy_{c}"], out[f"cloud_depth_{c}"] = cl[:, :2].copy(), cl[:, 2].copy()
        out[f"query_xy_{c}"], out[f"nn_index_{c}"] = qs[:, :2].copy(), qs[:, 2].astype(np.int64)
        out[f"nn_dist2_{c}"], out[f"nn_depth_{c}"] = qs[:, 3].copy(), qs[:, 4].copy()
    np.savez_compressed(os.path.join(HERE, name), **out)
    tot = sum(len(out[f"query_xy_{c}"]) for c in range(n_clouds))
    print("wrote", name + ":", n_clouds, "clouds,", tot, "queries")


if __name__ == "__main__":
    main()

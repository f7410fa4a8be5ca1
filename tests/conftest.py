"""pytest configuration: the `gpu` marker and shared fixture loaders."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The shared objects are build artefacts (git-ignored): on a fresh checkout build them before collection,
    # exactly as __graft_entry__.build() does (hipcc cross-compiles without a GPU).
    lib = os.path.join(ROOT, "groupnet_amd", "libgroupnet_hip.so")
    topk = os.path.join(ROOT, "oracle", "_build", "liboracle_topk.so")
    if not (os.path.exists(lib) and os.path.exists(topk)):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "groupnet_amd", "csrc")], check=True, stdout=subprocess.DEVNULL)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_state(name):
    """state_dict (reference key names) from tests/golden/weights_<name>.npz as CPU tensors."""
    with np.load(os.path.join(GOLDEN, f"weights_{name}.npz")) as z:
        return {k: torch.from_numpy(z[k].copy()) for k in z.files}


def load_case(name):
    with np.load(os.path.join(GOLDEN, f"case_{name}.npz")) as z:
        return {k: z[k].copy() for k in z.files}


def case_names():
    return sorted(os.path.basename(p)[len("case_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "case_*.npz")))


def weights_for(case):
    suffix = "_nmp2" if case.endswith("nmp2") else ""
    return load_state("pairwise" + suffix), load_state("hyper" + suffix), (2 if suffix else 1)


def uniforms(case, prefix):
    out, i = [], 0
    while f"{prefix}_U{i}" in case:
        out.append(torch.from_numpy(case[f"{prefix}_U{i}"]))
        i += 1
    return out

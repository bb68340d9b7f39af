"""-m gpu: A/B parity of the packer's compiler option.  The register packer's translation unit is built with the experimental
LLVM option -structurizecfg-skip-uniform-regions (kubernetes_autoscaler_amd/csrc/casim_pack_tu.hip: why, and what it was
caught doing to another kernel); tests/ab/libcasim_noskip.so is the same source without it.  A corpus of ~900 batches (every
fuzz family, both packers, fastpath, device-side subsets, C0..C4 at full size, tiled batches of simulations) must hash
identically through both builds — next to the oracle comparisons of the other GPU tests, which run the product build."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB_LIB = os.path.join(ROOT, "tests", "ab", "libcasim_noskip.so")


def _run(env_extra):
    env = dict(os.environ); env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "ab_corpus.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=1200)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_packer_results_do_not_depend_on_the_structurizer_option():
    # (always: a no-op when the library is newer than every source, and a stale A/B build would not even load after an ABI change)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "ab")], check=True)
    a = _run({})
    b = _run({"CASIM_LIB_PATH": AB_LIB})
    assert a["lib"] != b["lib"] and b["lib"] == AB_LIB
    assert a["batches"] == b["batches"] > 500
    assert a["sha256"] == b["sha256"], (a, b)

"""The Node.js side of the boundary: addon loads (CPU) and the drop-in JS modules match the goldens (GPU)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "pil2-stark-js_amd", "addon", "pil2gl.node")


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_addon_loads_and_exports():
    assert os.path.exists(ADDON), "addon not built (make -C pil2-stark-js_amd)"
    js = ("const m=require(%r);const a=m.native;"
          "for (const k of ['interpolate','fft','ifft','merkelize','merkelizeLevel','linearHashRows','poseidon','friFold',"
          "'friTranspose','buildXDev','buildZhInvDev','computeQSplitDev','xDivXSubXiDev','buildLevDev','computeEvalsDev','gprodDev','gsumDev','h1h2Dev','bn128Poseidon','bn128Merkelize','bn128MerkelizeDev','bn128LinearHashRows','bn128Convert','devAlloc','devFree','devUpload','devDownload','interpolateDev','merkelizeDev','groupProofDev','rootsFromGroupProofs','spongeAbsorb','bn128SpongeAbsorb','friVerifyFold','evalProgramDev'])"
          " if (typeof a[k] !== 'function') throw new Error('missing '+k);"
          "if (a.merkleNumNodes(256) !== 2044) throw new Error('merkleNumNodes');"
          "if (a.bn128MerkleNumNodes(256, 16) !== 273) throw new Error('bn128MerkleNumNodes');"
          "for (const k of ['fft','ifft','interpolate']) if (typeof m.fft_p[k] !== 'function') throw new Error(k);"
          "console.log('ok')") % os.path.join(ROOT, "pil2-stark-js_amd", "js", "index.js")
    out = subprocess.run([NODE, "-e", js], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_modules_match_goldens_on_gpu():
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "addon_parity.js")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "addon parity OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_whole_proof_driven_from_node():
    """prover.js's stage order over the JS drop-in modules: the proof equals the CPU checker's proof field by field"""
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "prove_flow.js")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "prove flow OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_config2_proof_from_node_device_resident():
    """config 2's shape proved from Node with HBM-resident buffers: digest of the proof equals the CPU checker's"""
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "prove_c2.js")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "prove c2 OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]

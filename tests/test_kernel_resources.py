"""Build hygiene for the hand-written conv kernels: no scratch memory, no register spills.  A kernel whose accumulators
or per-piece arrays fall out of the register file still computes the right values — the parity tests stay green — at a
fraction of the speed (round 4: one extra branch in a staging lambda of conv_halo_dma.hip put 640 bytes per lane into
scratch and cost 27 % of the step).  hipcc's -Rpass-analysis=kernel-resource-usage remarks are the check."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("source,kernels", [
    ("conv_halo_dma.hip", ["conv3x3_halo_dma_kernelILi2ELb1", "conv3x3_halo_dma_kernelILi2ELb0", "conv3x3_halo_dma_kernelILi1ELb0"]),
    ("conv_pipe.hip", ["conv3x3_pipe_kernelILi128ELi0", "conv3x3_pipe_kernelILi64ELi0", "conv3x3_pipe_kernelILi128ELi2"]),
    ("adam_pack.hip", ["adam_pack_kernel"]),
    ("conv_line64.hip", ["conv3x3_line64_kernelILi0", "conv3x3_line64_kernelILi2"]),
    ("conv_wgrad_pipe.hip", ["conv3x3_wgrad_pipe_kernelILi2ELi32ELb1", "conv3x3_wgrad_pipe_kernelILi2ELi32ELb0",
                             "conv3x3_wgrad_pipe_kernelILi1ELi32ELb1", "conv3x3_wgrad_pipe_kernelILi2ELi16ELb1",
                             "conv3x3_wgrad_pipe_kernelILi1ELi16ELb0"]),
    ("conv_wgrad_tr.hip", ["conv_wgrad_tr_dma_kernel", "conv_wgrad_tr_kernelILi2ELi2ELb0"]),
    ("conv_gather.hip", ["conv_gather_line_kernel"]),
    ("conv_s2d_dgrad.hip", ["conv_s2d_dgrad_kernelILi32ELi4", "conv_s2d_dgrad_kernelILi16ELi4", "conv_s2d_dgrad_kernelILi32ELi8",
                            "conv_s2d_dgrad_kernelILi16ELi8"]),
    ("conv_s2d_fwd.hip", ["conv_s2d_fwd_kernelILi32ELi128ELi4ELi4", "conv_s2d_fwd_kernelILi16ELi128ELi4ELi4",
                          "conv_s2d_fwd_kernelILi32ELi128ELi8ELi2", "conv_s2d_fwd_kernelILi16ELi128ELi8ELi2",
                          "conv_s2d_fwd_kernelILi32ELi256ELi8ELi4", "conv_s2d_fwd_kernelILi16ELi256ELi8ELi4",
                          "conv_s2d_fwd_kernelILi32ELi128ELi8ELi4", "conv_s2d_fwd_kernelILi32ELi64ELi4ELi2", "conv_s2d_fwd_kernelILi16ELi64ELi4ELi2"]),
])
def test_hot_kernels_use_no_scratch(tmp_path, source, kernels):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, source), "-o",
                          str(tmp_path / "k.o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", out.stderr)[1:]
    seen = {}
    for blk in blocks:
        name = blk.split()[0]
        scratch = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk)
        spill = re.search(r"VGPRs Spill: (\d+)", blk)
        seen[name] = (int(scratch.group(1)) if scratch else None, int(spill.group(1)) if spill else None)
    for k in kernels:
        hits = {n: v for n, v in seen.items() if k in n}
        assert hits, (k, sorted(seen))
        for n, (scratch, spill) in hits.items():
            assert scratch == 0 and spill == 0, (n, "scratch bytes/lane", scratch, "spilled VGPRs", spill)


def _makefile_flags(stem):
    """The per-file extra flags of csrc/Makefile (CXXFLAGS_<stem> = ...)."""
    for line in open(os.path.join(CSRC, "Makefile")):
        m = re.match(r"CXXFLAGS_%s\s*=\s*(.*)" % re.escape(stem), line)
        if m:
            return m.group(1).split()
    return []


def _crossed_packed_fp32(asm_path):
    """(kernel, instruction) of every packed fp32 instruction with a CROSSED source: op_sel = 1 and op_sel_hi = 0 for the
    same operand, i.e. the low result reads the high register of the pair and the high result the low one."""
    hits, cur = [], None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        if not re.match(r"\s*v_pk_\w+_f32\s", line):
            continue
        a = re.search(r"op_sel:\[([01,]+)\]", line)
        b = re.search(r"op_sel_hi:\[([01,]+)\]", line)
        osel = [int(v) for v in a.group(1).split(",")] if a else [0, 0, 0]
        ohi = [int(v) for v in b.group(1).split(",")] if b else [1, 1, 1]
        if any(osel[i] == 1 and ohi[i] == 0 for i in range(min(len(osel), len(ohi)))):
            hits.append((cur, line.strip()))
    return hits


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason="needs hipcc")
def test_no_crossed_packed_fp32_operands(tmp_path):
    """Round 6 (profiles/r06_torgb_contention.txt): `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` — what hipcc's SLP
    vectoriser made of the to-RGB data gradient — lost the low product in lanes 48-63 whenever waves of another process (or
    stream) shared the GPU.  torgb.hip is built without the SLP vectoriser since; this test compiles EVERY translation unit
    with the Makefile's flags and fails on any packed fp32 instruction with a crossed source.  (Two more kernels had one — a
    crossed `v_pk_fma_f32` operand in the 8-channel RGB upsample variant, which nothing launched and which is gone, and in
    `resize_norm_fwd_kernel`, whose two rows are scalar fmas now; neither ever differed in the contention probe.)"""
    from concurrent.futures import ThreadPoolExecutor

    known = {}
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))

    def compile_one(src):
        out = str(tmp_path / (src + ".s"))
        r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"]
                           + _makefile_flags(src[:-4]) + ["-S", "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", out],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return src, _crossed_packed_fp32(out)

    with ThreadPoolExecutor(max_workers=4) as pool:
        found = dict(pool.map(compile_one, srcs))
    assert found["torgb.hip"] == [], found["torgb.hip"]
    assert "-fno-slp-vectorize" in _makefile_flags("torgb")
    seen = {}
    for src, hits in found.items():
        for kernel, ins in hits:
            key = next((k for k in known if k in (kernel or "")), None)
            assert key is not None, "new crossed packed-fp32 operand in %s, kernel %s: %s" % (src, kernel, ins)
            seen[key] = seen.get(key, 0) + 1
    for key, n in seen.items():
        assert n <= known[key], (key, n)

"""Race detector for the multi-stream step at FULL size: two fresh Trainers from the same seeds run N steps of the
bench workload (256 px, B=32, GAE=2, bf16, HIP streams on); every loss scalar and a parameter checksum must be
bit-identical.  All our reductions are fixed-order, so a missing stream dependency shows up as run-to-run noise.
The frozen classifier / LPIPS run on MIOpen, whose default algorithms are NOT run-to-run reproducible (measured:
identical input, logits differing at 1e-7, amplified to 2e-3 in D's bf16 output) — the check pins them with
torch.backends.cudnn.deterministic.      python tools/determinism_check.py [steps] [image_size] [batch]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(steps=5, image_size=256, batch=32, lazy=False, trainers=2):
    """lazy=True: no loss scalar is read until the last step (reading one waits for the step's device->host copy), so
    the host runs ahead of the GPU by whole steps exactly as it does in real training / bench.py."""
    import torch

    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
    import hip_backend as hb
    import ops

    hb.load_library()
    ops.set_precision(os.environ.get("DET_PRECISION", "bf16"))
    prev_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    if os.environ.get("DET_TORCH_WARN") == "1":  # name every ATen op of the step that is flagged as not reproducible
        torch.use_deterministic_algorithms(True, warn_only=True)
    # DET_CLASSIFIER / DET_PL_EVERY / DET_START_STEP: BASELINE config 4's variant (MobileNetV2, path-length steps, which start
    # after step 5000) through the same check
    start = int(os.environ.get("DET_START_STEP", "0"))
    a = argparse.Namespace(batch=batch, image_size=image_size, gae=2, classifier=os.environ.get("DET_CLASSIFIER", "resnet"),
                           pl_every=int(os.environ.get("DET_PL_EVERY", "32")), workdir="/tmp/sb_det",
                           precision=os.environ.get("DET_PRECISION", "bf16"), device_rng=int(os.environ.get("DET_DEVICE_RNG", "0")))
    runs = []
    try:
        if os.environ.get("DET_WARM", "1") != "0":
            # A throw-away Trainer first.  The FIRST Trainer of a process differs from every later one deterministically
            # (same value in every process; later Trainers agree with each other bit for bit): the autograd engine orders
            # ready nodes by per-thread sequence numbers, and the nodes the double backward of a penalty step creates on
            # the engine's worker thread start from that thread's counter, so in a fresh process they interleave
            # differently with the main thread's nodes — a different accumulation ORDER of the parameter gradients that
            # receive several contributions (profiles/probes/first_diff_probe.py: the call sequences differ; every kernel
            # output agrees until they do).  Not a race: the check is about run-to-run noise in steady state.
            bench.seed_all(42)
            tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
            tr.steps = start
            tr.train()
            torch.cuda.synchronize()
            del tr
            torch.cuda.empty_cache()
        for _ in range(trainers):
            bench.seed_all(42)
            tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
            tr.steps = start
            rows = []
            for _i in range(steps):
                tr.train()
                if not lazy or _i == steps - 1:
                    rows.append((tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss, tr.last_gp_loss))
            chk = float(sum(p.detach().double().abs().sum() for p in tr.StylEx.parameters()))
            runs.append((rows, chk))
            del tr
            torch.cuda.empty_cache()
    finally:
        torch.backends.cudnn.deterministic = prev_det
        ops.set_precision("fp32")
    return runs


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    os.chdir(ROOT)
    r = run(steps, size, batch=int(sys.argv[3]) if len(sys.argv) > 3 else 32, trainers=int(os.environ.get("DET_TRAINERS", "2")))
    if len(r) > 2:
        print("all parameter checksums", [x[1] for x in r])
    for i, (x, y) in enumerate(zip(r[0][0], r[1][0])):
        print(i, "OK  " if x == y else "DIFF", x, y if x != y else "")
    print("parameter checksum", r[0][1], r[1][1])
    print("bit-identical:", r[0] == r[1])
    sys.exit(0 if r[0] == r[1] else 1)

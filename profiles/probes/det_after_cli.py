"""Reproduce: the 128 px determinism check run AFTER the CLI end-to-end test in the same process."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), ROOT]
os.chdir(ROOT)
import test_hip_cli_gpu as t
import determinism_check
if os.environ.get("SKIP_CLI") != "1":
    t.test_train_from_folder_on_gpu_bf16(pathlib.Path(tempfile.mkdtemp()))
runs = determinism_check.run(steps=5, image_size=128, batch=16, trainers=int(os.environ.get("DET_TRAINERS", "3")))
for i in range(5):
    print(i, [r[0][i] for r in runs])
print([r[1] for r in runs])

import sys, os, numpy as np, importlib.util
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
spec = importlib.util.spec_from_file_location("curve_check", os.path.join(ROOT,"tools","curve_check.py"))
cc = importlib.util.module_from_spec(spec); spec.loader.exec_module(cc)
got, ref = cc.run(n=12, precision="fp32")
err = np.abs(got[:, :4] - ref[:, :4]) / np.maximum(1.0, np.abs(ref[:, :4]))
print("HIP fp32 vs ref8:", np.array2string(err.max(1), precision=2))
np.save(os.path.join(ROOT,"gpurun_out","r03c_curve12_hip.npy"), got)

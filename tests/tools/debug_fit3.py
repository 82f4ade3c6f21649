"""The value sums of fit_value_kernel3 against fit_accumulate_kernel2<0> (FRI_HIP_K4_VALUE3=0) for one shape: prints the differing entries. GPU only."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np

    import frave_amd
    from tests.common import gen_image

    w, h, kind = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    P = frave_amd.Plan(frave_amd.Context(0), w, h, 1)
    co = P.transform_quant(gen_image(kind, w, h, 1, 3))
    g = P.fit_value_sums(co, 0)
    np.save(sys.argv[5], g)
    sys.exit(0)
import numpy as np

w, h, kind = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (512, 384, "noise")
out = {}
for v3 in ("0", "1"):
    env = dict(os.environ, FRI_HIP_TUNING="1", FRI_HIP_K4_VALUE3=v3)
    f = f"/tmp/fit3_{v3}.npy"
    subprocess.run([sys.executable, __file__, "child", str(w), str(h), kind, f], check=True, env=env)
    out[v3] = np.load(f)
a, b = out["0"], out["1"]
print("equal:", np.array_equal(a, b))
for g in range(3):
    d = b[g] - a[g]
    print(f"group {g}: old diag {np.diag(a[g]).tolist()}")
    print(f"         new diag {np.diag(b[g]).tolist()}")
    if d.any():
        print("  ratio new/old on the diagonal:", (np.diag(b[g]) / np.maximum(np.diag(a[g]), 1)).round(4).tolist())

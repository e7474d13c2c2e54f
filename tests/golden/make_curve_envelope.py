"""tests/golden/train_curve_envelope.npz: the CPU PORT of the trainer (lav_amd/train/lav.py on torch CPU ops, tools/curve_cpu.py) run
over the 500 steps of tests/golden/train_curve.npz with different thread counts - samples of how far a second float32 implementation
of the same optimisation drifts from the reference's single CPU run (an envelope instead of widened bars).

    for t in 8 6 5 7 2 1; do python tools/curve_cpu.py --threads $t --steps 500 --out gpurun_out/r5/curve_cpu_t$t.npy; done   (hours)
    python tests/golden/make_curve_envelope.py gpurun_out/r5/curve_cpu_t*.npy [profiles/r04_loss_curves.npz]
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
ref = np.load(os.path.join(here, "train_curve.npz"))
curves, names = [], []
for path in sys.argv[1:]:
    if path.endswith(".npz"):      # round 4's two samples
        z = np.load(path)
        for k in ("cpu_t3", "cpu_t4"):
            if k in z.files and len(z[k]) == 500:
                curves.append(z[k]); names.append(k)
    else:
        a = np.load(path)
        if len(a) == 500:
            curves.append(a); names.append("cpu_" + os.path.basename(path).split("_")[-1][:-4])
order = np.argsort(names)
np.savez_compressed(os.path.join(here, "train_curve_envelope.npz"), curves=np.array([curves[i] for i in order], np.float64),
                    names=np.array([names[i] for i in order]), keys=ref["keys"])
print("envelope of", [names[i] for i in order])

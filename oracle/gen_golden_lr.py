"""Golden trajectories of the learning-rate schedule from the reference's own class (learning_rates.py imports cleanly:
torch only).      python oracle/gen_golden_lr.py      (build container only: reads /root/reference)
Test infrastructure: writes tests/golden/learning_rates.npz."""
import contextlib
import importlib.util
import io
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Opt:
    def __init__(self):
        self.param_groups = [{'lr': 0.0}, {'lr': 0.0}]


def main():
    spec = importlib.util.spec_from_file_location("ref_lr", "/root/reference/learning_rates.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    for style in ("linear", "cosine", "constant"):
        o = _Opt()
        with contextlib.redirect_stdout(io.StringIO()):
            s = m.AnnealingLR(o, 3e-4, 50, 400, decay_style=style, decay_ratio=0.1)
        lrs = []
        for _ in range(480):
            s.step()
            lrs.append(o.param_groups[0]['lr'])
        out[style] = np.array(lrs)
        out[style + "_sd_num_iters"] = np.array(s.state_dict()['num_iters'])
        out[style + "_decay_ratio"] = np.array(s.state_dict()['decay_ratio'])
    np.savez(os.path.join(ROOT, "tests", "golden", "learning_rates.npz"), **out)


if __name__ == "__main__":
    main()

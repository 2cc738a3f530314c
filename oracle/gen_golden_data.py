"""Golden vectors for the data readers from the reference's own classes, taken out of their
modules with `ast` (the modules import lmdb / torchvision / the sentencepiece tokenizer, none of which is needed here).
    python oracle/gen_golden_data.py        (build container only: reads /root/reference)
Test infrastructure: writes tests/golden/data_utils.npz."""
import ast
import os
import random

import numpy as np
from torch.utils import data
from torch.utils.data import Dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def take(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def main():
    ns = {"np": np, "random": random, "data": data, "Dataset": Dataset, "os": os}
    RandomMappingDataset, = take("/root/reference/data_utils/configure_data.py", ["RandomMappingDataset"], ns)
    BinaryDataset, = take("/root/reference/data_utils/datasets.py", ["BinaryDataset"], ns)
    rs = np.random.RandomState(5)
    rows = np.full((7, 64 + 1024), -1, dtype=np.int32)
    for i in range(7):
        n = rs.randint(3, 40)
        rows[i, :n] = rs.randint(8192, 58192, n)
        rows[i, 64:] = rs.randint(0, 8192, 1024)
    path = "/tmp/_golden_rows.bin"
    rows.tofile(path)
    ds = BinaryDataset(path, lambda r: np.array(r))
    out = {"rows": rows, "read_back": np.stack([ds[i] for i in range(len(ds))]), "n": np.array(len(ds))}
    rm = RandomMappingDataset(list(range(1000)))
    out["mapping"] = np.array([rm[i] for i in range(64)])
    out["mapping_len"] = np.array(len(rm))
    np.savez(os.path.join(ROOT, "tests", "golden", "data_utils.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

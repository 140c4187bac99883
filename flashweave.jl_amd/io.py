"""On-disk formats either side of the path: the reference's `.edgelist` network format
(src/io.jl:338-389: two header lines `# header\\t<ids>` / `# meta mask\\t<bools>`, then `id<TAB>id<TAB>weight`) and
delimited OTU tables (src/io.jl:155-191: first row = variable ids, first column = sample ids)."""
import numpy as np


def write_edgelist(path, edges, header=None, meta_mask=None, n_vars=None):
    """edges: {(i, j): weight} with 0-based variable indices.  Edge order follows the reference's writer
    (upper triangle of the CSC adjacency: by larger endpoint, then smaller; src/io.jl:347-356)."""
    if n_vars is None:
        n_vars = (max(max(e) for e in edges) + 1) if edges else 0
    if header is None:
        header = ["X%d" % (i + 1) for i in range(n_vars)]
    if meta_mask is None:
        meta_mask = [False] * len(header)
    with open(path, "w") as f:
        f.write("# header\t" + ",".join(header) + "\n")
        f.write("# meta mask\t" + ",".join("true" if m else "false" for m in meta_mask) + "\n")
        for (i, j), w in sorted(edges.items(), key=lambda kv: (max(kv[0]), min(kv[0]))):
            a, b = (i, j) if i < j else (j, i)
            f.write("%s\t%s\t%r\n" % (header[a], header[b], float(w)))


def read_edgelist(path):
    """-> (edges {(i, j): w} with i < j, header, meta_mask)"""
    with open(path) as f:
        lines = f.read().rstrip("\n").split("\n")
    header = lines[0].split("\t")[-1].split(",")
    meta_mask = [x == "true" for x in lines[1].split("\t")[-1].split(",")]
    inv = {h: i for i, h in enumerate(header)}
    edges = {}
    for line in lines[2:]:
        if not line:
            continue
        a, b, w = line.split("\t")
        i, j = inv[a], inv[b]
        edges[(min(i, j), max(i, j))] = float(w)
    return edges, header, meta_mask


def read_table(path, delimiter=None):
    """Delimited count table with a header row and an id column (src/io.jl:155-191) -> (counts float64, header, row ids)."""
    if delimiter is None:
        delimiter = "," if path.endswith(".csv") else "\t"
    with open(path) as f:
        rows = [r.rstrip("\n").split(delimiter) for r in f if r.strip()]
    header = rows[0][1:]
    ids = [r[0] for r in rows[1:]]
    counts = np.array([[float(x) for x in r[1:]] for r in rows[1:]], dtype=np.float64)
    return counts, header, ids

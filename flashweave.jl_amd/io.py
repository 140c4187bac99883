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


# ---- load_data (src/io.jl:28-118,147-246): delimited tables with or without a row-id column, transposed tables, BIOM (JSON) ------
def _parse_cell(x):
    try:
        return float(x)
    except ValueError:
        return x


def _has_row_ids(rows, header):
    """hasrowids (src/io.jl:149-150): an empty first header cell, or a first column of pairwise different strings."""
    first = [r[0] for r in rows]
    return header[0] == "" or (len(set(first)) == len(rows) and isinstance(first[0], str))


def load_dlm(data_path, meta_path=None, transposed=False, type_data=True):
    """load_dlm (src/io.jl:152-191) -> (data, header, meta_data, meta_header).  data: Float64 matrix (type_data) or a list of rows of
    mixed cells (meta data: numbers stay numbers, factors stay strings)."""
    sep = "\t" if data_path.endswith(".tsv") else ","
    with open(data_path) as f:
        raw = [[_parse_cell(c) for c in line.rstrip("\n").rstrip("\r").split(sep)] for line in f if line.strip()]
    if transposed:
        raw = [list(col) for col in zip(*raw)]
    header, rows = raw[0], raw[1:]
    if _has_row_ids(rows, header):
        rows = [r[1:] for r in rows]
        header = header[1:]
    header = [("%g" % h if isinstance(h, float) and h == int(h) else str(h)) for h in header]   # readdlm types numeric ids: "1994.0" -> "1994"
    data = np.array(rows, dtype=np.float64) if type_data else rows
    meta_data = meta_header = None
    if meta_path is not None:
        meta_data, meta_header, _, _ = load_dlm(meta_path, transposed=transposed, type_data=False)
    return data, header, meta_data, meta_header


def load_biom_json(data_path):
    """load_biom_json (src/io.jl:194-206): BIOM 1.0 (JSON), dense or sparse -> (samples x OTUs counts, OTU ids)."""
    import json
    with open(data_path) as f:
        js = json.load(f)
    n_obs, n_samp = js["shape"]
    if js["matrix_type"] == "sparse":
        tab = np.zeros((n_obs, n_samp), dtype=np.int64)
        for i, j, v in js["data"]:
            tab[int(i), int(j)] = int(v)
    else:
        tab = np.array(js["data"], dtype=np.int64)
    return tab.T.astype(np.float64), [r["id"] for r in js["rows"]]


def load_data(data_path, meta_path=None, transposed=False):
    """load_data (src/io.jl:28-66) for the formats readable without extra packages: .tsv / .csv tables and BIOM 1.0 (JSON).
    BIOM 2.x (HDF5) and JLD2 need libraries this environment does not have and raise."""
    ext = data_path.rsplit(".", 1)[-1].lower()
    if ext in ("tsv", "csv"):
        return load_dlm(data_path, meta_path, transposed=transposed)
    if ext == "biom":
        with open(data_path, "rb") as f:
            if f.read(8) == b"\x89HDF\r\n\x1a\n":
                raise NotImplementedError("BIOM 2.x (HDF5) needs h5py, which is not installed; convert with `biom convert --to-json`")
        data, header = load_biom_json(data_path)
        meta_data = meta_header = None
        if meta_path is not None:
            meta_data, meta_header, _, _ = load_dlm(meta_path, type_data=False)
        return data, header, meta_data, meta_header
    raise ValueError("load_data: unsupported format .%s (supported: .tsv, .csv, .biom as JSON)" % ext)


# ---- GML networks (src/io.jl:392-482) -------------------------------------------------------------------------------------------
def write_gml(path, edges, header=None, meta_mask=None, n_vars=None):
    """write_gml (src/io.jl:392-423): undirected graph, 1-based node ids, node attributes `label` and `mv` (meta variable), edge
    attribute `weight`; edges in the order of the reference's graph iterator (by smaller endpoint, then larger)."""
    if n_vars is None:
        n_vars = len(header) if header is not None else ((max(max(e) for e in edges) + 1) if edges else 0)
    if header is None:
        header = ["X%d" % (i + 1) for i in range(n_vars)]
    if meta_mask is None:
        meta_mask = [False] * len(header)
    with open(path, "w") as f:
        f.write("graph [\n\tdirected 0\n")
        for i, h in enumerate(header):
            f.write("\tnode [\n\t\tid %d\n\t\tlabel \"%s\"\n\t\tmv %d\n\t]\n" % (i + 1, h, int(bool(meta_mask[i]))))
        for (i, j), w in sorted(((min(e), max(e)), w) for e, w in edges.items()):
            f.write("\tedge [\n\t\tsource %d\n\t\ttarget %d\n\t\tweight %r\n\t]\n" % (i + 1, j + 1, float(w)))
        f.write("]\n")


def read_gml(path):
    """read_gml (src/io.jl:444-482) -> (edges {(i, j): w} with i < j, 0-based; header; meta_mask)."""
    nodes, edges, cur, kind = {}, {}, None, None
    with open(path) as f:
        for line in f:
            t = line.strip()
            if t.startswith("node") or t.startswith("edge"):
                kind, cur = t.split()[0], {}
            elif t.startswith("]") and cur is not None:
                if kind == "node":
                    nodes[int(cur["id"])] = (cur["label"].strip('"'), bool(int(cur.get("mv", "0"))))
                else:
                    a, b = int(cur["source"]) - 1, int(cur["target"]) - 1
                    edges[(min(a, b), max(a, b))] = float(cur["weight"])
                cur = None
            elif cur is not None and t:
                k, v = t.split(None, 1)
                cur[k] = v
    n = max(nodes) if nodes else 0
    header = [nodes.get(i + 1, ("", False))[0] for i in range(n)]
    meta_mask = [nodes.get(i + 1, ("", False))[1] for i in range(n)]
    return edges, header, meta_mask


def save_network(path, edges, header=None, meta_mask=None):
    """save_network (src/io.jl:300-336) by extension: .edgelist or .gml."""
    if path.endswith(".gml"):
        write_gml(path, edges, header, meta_mask)
    elif path.endswith(".edgelist"):
        write_edgelist(path, edges, header, meta_mask)
    else:
        raise ValueError("save_network: unsupported format (supported: .edgelist, .gml)")


def load_network(path):
    return read_gml(path) if path.endswith(".gml") else read_edgelist(path)

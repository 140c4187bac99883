"""Formats of SURVEY 8f-4: edgelist round trip (reference test/io.jl:29-43) against the golden files, table reader."""
import os
import tempfile

from flashweave_jl_amd import io as fio
from tests.util import GOLDEN, read_edgelist


def test_edgelist_roundtrip_against_golden():
    for name in ("exp_fz_maxk3", "exp_mi_maxk0", "exp_mi_nz_maxk3"):
        path = "%s/learning_expected/%s.edgelist" % (GOLDEN, name)
        edges, header, mask = fio.read_edgelist(path)
        assert edges == read_edgelist(path) and len(header) == 50 and not any(mask)
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "net.edgelist")
            fio.write_edgelist(out, edges, header, mask)
            assert open(out).read().strip() == open(path).read().strip()  # byte-identical incl. edge order


def test_read_table():
    counts, header, ids = fio.read_table(GOLDEN + "/HMP_SRA_gut_small.tsv")
    assert counts.shape == (351, 50) and len(header) == 50 and len(ids) == 351 and counts[0, 0] == 141.0


def test_load_data_formats_agree():
    # reference test/io.jl:84-143: the tiny table in every supported format loads to the same matrix and header
    base = GOLDEN + "/HMP_SRA_gut_tiny"
    data, header, _, _ = fio.load_data(base + ".tsv")
    assert data.shape == (19, 19) and header[0] == "BG1994" and data[0, 0] == 141.0 and data[1, 1] == 15.0
    for suff, kw in (("_ids.tsv", {}), (".csv", {}), ("_json.biom", {}), ("_ids_transposed.tsv", dict(transposed=True))):
        d2, h2, _, _ = fio.load_data(base + suff, **kw)
        assert (d2 == data).all() and h2 == header, suff
    d3, h3, _, _ = fio.load_data(base + "_numIDs.tsv")       # numeric ids ("numeric IDs" test): BG1994 -> "1994"
    assert (d3 == data).all() and h3 == [h[2:] for h in header]
    # string factors in the meta table stay strings, numbers stay numbers (test "string factors")
    _, _, md, mh = fio.load_data(base + "_ids.tsv", GOLDEN + "/HMP_SRA_gut_tiny_meta_oneHotTest.tsv")
    assert len(md) == 19 and len(mh) == len(md[0]) and any(isinstance(v, str) for v in md[0])


def test_gml_roundtrip():
    # reference test/io.jl:29-43: save_network / load_network give back the same graph, for both text formats
    edges, header, mask = fio.read_edgelist(GOLDEN + "/learning_expected/exp_fz_maxk3.edgelist")
    mask[3] = True
    with tempfile.TemporaryDirectory() as d:
        for ext in ("gml", "edgelist"):
            out = os.path.join(d, "net." + ext)
            fio.save_network(out, edges, header, mask)
            e2, h2, m2 = fio.load_network(out)
            assert e2 == edges and h2 == header and m2 == mask, ext
        txt = open(os.path.join(d, "net.gml")).read()
        assert txt.startswith("graph [\n\tdirected 0\n\tnode [\n\t\tid 1\n\t\tlabel \"%s\"\n\t\tmv 0\n\t]" % header[0])

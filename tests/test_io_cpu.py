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

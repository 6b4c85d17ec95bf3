"""Stand-in stages for relate_amd.dist (tests/test_dist_cpu.py::test_launcher_really_starts_two_ranks): the interface
of relate_amd.api's stage functions, each call recorded as a line `<rank> <stage> <chunk> [sections]` in
<out_dir>/calls_rank<rank>.txt -- no GPU, no library."""
import os


def _log(out_dir, what):
    rank = os.environ.get("RANK", "0")
    with open(os.path.join(out_dir, "calls_rank%s.txt" % rank), "a") as f:
        f.write("%s %s\n" % (rank, what))


def num_sections(out_dir, chunk_index=0):
    return 3 + chunk_index


def stage_paint(out_dir, chunk_index=0, painting=None, device=0, **kw):
    _log(out_dir, "paint %d dev%d" % (chunk_index, device))


def stage_build_topology(out_dir, chunk_index, first_section, last_section, painting=None, device=0, **kw):
    _log(out_dir, "build %d %d-%d dev%d" % (chunk_index, first_section, last_section, device))


def stage_paint_build_topology(out_dir, chunk_index, first_section, last_section, painting=None, device=0, **kw):
    _log(out_dir, "fused %d %d-%d dev%d" % (chunk_index, first_section, last_section, device))


def stage_find_equivalent_branches(out_dir, chunk_index=0):
    _log(out_dir, "feb %d" % chunk_index)

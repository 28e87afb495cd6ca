"""Host side of the specialised code objects (raisimlib_amd/csrc/rsb_spec.hip; no GPU): manifest lines, file names, what the compile entry point refuses, and that
build() left one code object per manifest line in rsb_spec_dir()."""
import os
import re

import pytest

from raisimlib_amd import _capi, build as _b

C = _capi.C


def _name(line):
    buf = C.create_string_buffer(256)
    rc = _capi.lib().rsb_spec_file_name(line.encode(), buf, 256)
    return rc, buf.value.decode()


def _lines():
    return [l.strip() for l in open(_b.SPEC_MANIFEST) if l.strip() and not l.startswith("#")]


def test_manifest_lines_name_one_code_object_each_and_build_left_them_there(built_lib):
    L = _capi.lib()
    d = L.rsb_spec_dir().decode()
    names = set()
    for line in _lines():
        rc, n = _name(line)
        assert rc == 0 and re.fullmatch(r"step_\d+_\d+_\d+_\d+p?_[0-9a-f]{16}\.hsaco", n), (line, n)
        names.add(n)
        assert os.path.getsize(os.path.join(d, n)) > 10000, f"{n}: not built (python -c 'import __graft_entry__ as g; g.build()')"
    assert len(names) == len(_lines()) >= 11                       # distinct keys -> distinct files
    assert names <= set(os.listdir(d))                               # (beside them: objects a GPU session compiled on demand; build() removes those of older sources)
    # every field of step_spec.h appears in every line, in the list's order
    fields = re.findall(r"X\((\w+),", open(os.path.join(_b.CSRC, "step_spec.h")).read().split("#define RSB_SPEC_FIELDS(X)")[1].split("#ifdef")[0])
    for line in _lines():
        assert re.findall(r"-DRSB_SPEC_(\w+)=", line) == fields, line


def test_the_file_name_follows_the_key_and_the_source_hash(built_lib):
    line = _lines()[0]
    rc, a = _name(line)
    rc2, b = _name(line.replace("-DRSB_SPEC_NSUB=4", "-DRSB_SPEC_NSUB=2"))
    rc3, c = _name(line.replace(" |", " p |", 1))                   # the profiling twin of the same key
    assert rc == rc2 == rc3 == 0 and len({a, b, c}) == 3 and c.split("_")[4].endswith("p")
    assert _name(line)[1] == a                                      # deterministic


@pytest.mark.parametrize("bad", ["", "16 8 0 4", "16 8 0 | -DRSB_SPECIALIZED", "16 8 0 4 | -DFOO=1", "16 8 0 4 | -DRSB_SPECIALIZED; rm -rf /tmp/x",
                                 "16 8 0 4 | -DRSB_SPECIALIZED -DRSB_SPEC_NB=$(id)", "16 8 0 4 | -DRSB_SPECIALIZED -DRSB_SPEC_NB='13'", "x y z w | -DRSB_SPECIALIZED"])
def test_malformed_lines_are_refused_before_anything_reaches_a_shell(built_lib, bad):
    L = _capi.lib()
    assert L.rsb_spec_compile(bad.encode()) < 0
    assert _name(bad)[0] < 0

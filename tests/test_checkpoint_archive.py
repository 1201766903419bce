"""(f)-3: the `*.pt.tch` container (libtorch named-tensor archive) under Agent::save_params / load_params
(border-tch-agent/src/dqn/base.rs:348-362 -> tch `VarStore::{save,load}` -> libtorch OutputArchive / torch::jit::load).

Host-only: the container code is exercised through `bdr_checkpoint_write` / `bdr_checkpoint_read`; no GPU involved.
Pinning: fixtures written by libtorch itself (tests/golden/make_archive_fixtures.sh), and libtorch (torch.jit.load here,
the C++ InputArchive path in oracle/libtorch_archive.cpp when it has been built) reading what the library writes.
"""
import os
import subprocess
import zipfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
TOOL = os.path.join(HERE, "..", "oracle", "_build", "libtorch_archive")


def fill(index, dims):
    n = int(np.prod(dims, dtype=np.int64))
    return np.sin(0.37 * (np.arange(n, dtype=np.float64) + 1) + index).astype(np.float32).reshape(dims)


MLP_SPEC = [("mlp.ln0.weight", (64, 4)), ("mlp.ln0.bias", (64,)), ("mlp.ln1.weight", (64, 64)), ("mlp.ln1.bias", (64,)),
            ("mlp.ln2.weight", (2, 64)), ("mlp.ln2.bias", (2,))]
CNN_SPEC = [("c1.weight", (3, 4, 8, 8)), ("c1.bias", (3,)), ("l2.weight", (6, 5)), ("l2.bias", (6,)), ("log_alpha", (1,))]


@pytest.fixture(scope="module")
def ck():
    from border_amd import checkpoint
    return checkpoint


@pytest.mark.parametrize("fname,spec", [("libtorch_mlp_qnet.pt.tch", MLP_SPEC), ("libtorch_small_cnn.pt.tch", CNN_SPEC)])
def test_reads_archives_written_by_libtorch(ck, fname, spec):
    got = ck.read(os.path.join(GOLD, fname), spec)
    for i, (name, dims) in enumerate(spec):
        assert got[name].shape == dims
        assert np.array_equal(got[name], fill(i, dims)), name
    # a subset, in another order: matching is by name
    sub = [spec[3], spec[0]]
    got = ck.read(os.path.join(GOLD, fname), sub)
    assert np.array_equal(got[spec[0][0]], fill(0, spec[0][1])) and np.array_equal(got[spec[3][0]], fill(3, spec[3][1]))


def test_read_errors(ck, tmp_path):
    from border_amd import BdrError
    f = os.path.join(GOLD, "libtorch_mlp_qnet.pt.tch")
    with pytest.raises(BdrError, match="missing"):
        ck.read(f, [("mlp.ln3.weight", (2, 64))])
    with pytest.raises(BdrError, match="shape"):
        ck.read(f, [("mlp.ln2.bias", (3,))])
    with pytest.raises(BdrError):
        ck.read(str(tmp_path / "absent.pt.tch"), MLP_SPEC)
    junk = tmp_path / "junk.pt.tch"
    junk.write_bytes(b"not a zip archive at all, but longer than an end record")
    with pytest.raises(BdrError, match="zip"):
        ck.read(str(junk), MLP_SPEC)
    # truncated archive
    data = open(f, "rb").read()
    cut = tmp_path / "cut.pt.tch"
    cut.write_bytes(data[:len(data) // 2])
    with pytest.raises(BdrError):
        ck.read(str(cut), MLP_SPEC)


def test_libtorch_reads_what_the_library_writes(ck, tmp_path):
    import torch
    tensors = {name: fill(10 + i, dims) for i, (name, dims) in enumerate(MLP_SPEC + [("c1.weight", (32, 4, 8, 8)), ("log_alpha", (1,))])}
    path = str(tmp_path / "qnet.pt.tch")
    ck.write(path, tensors)
    # container level: a plain, uncompressed zip whose directory is named after the file stem, tensor data 64-byte aligned
    z = zipfile.ZipFile(path)
    assert z.testzip() is None
    names = z.namelist()
    assert "qnet.pt/data.pkl" in names and "qnet.pt/version" in names and "qnet.pt/code/__torch__.py" in names
    raw = open(path, "rb").read()
    for info in z.infolist():
        assert info.compress_type == zipfile.ZIP_STORED
        if "/data/" in info.filename:
            name_len, extra_len = int.from_bytes(raw[info.header_offset + 26:info.header_offset + 28], "little"), \
                int.from_bytes(raw[info.header_offset + 28:info.header_offset + 30], "little")
            assert (info.header_offset + 30 + name_len + extra_len) % 64 == 0
    # libtorch's TorchScript loader (what tch's VarStore::load calls): same names, order, shapes, bytes
    m = torch.jit.load(path)
    got = list(m.named_parameters())
    assert [n for n, _ in got] == list(tensors)
    for n, p in got:
        assert p.dtype == torch.float32 and tuple(p.shape) == tensors[n].shape
        assert np.array_equal(p.detach().numpy(), tensors[n]), n
    # and back through the library's own reader
    back = ck.read(path, [(k, v.shape) for k, v in tensors.items()])
    assert all(np.array_equal(back[k], v) for k, v in tensors.items())


def test_many_tensors_cross_the_one_byte_memo_limit(ck, tmp_path):
    """> 256 pickle memo slots (LONG_BINPUT / LONG_BINGET) and 2- and 4-byte integer encodings."""
    import torch
    tensors = {f"mlp.ln{i}.weight": fill(i, (1 + i % 3, 300 if i == 7 else 2)) for i in range(120)}
    tensors["big"] = fill(500, (70000,))
    path = str(tmp_path / "many.pt.tch")
    ck.write(path, tensors)
    got = dict(torch.jit.load(path).named_parameters())
    assert list(got) == list(tensors)
    assert all(np.array_equal(got[k].detach().numpy(), v) for k, v in tensors.items())
    back = ck.read(path, [(k, v.shape) for k, v in tensors.items()])
    assert all(np.array_equal(back[k], v) for k, v in tensors.items())


def test_reads_a_scripted_python_module_with_submodules(ck, tmp_path):
    """A Nature-CNN-shaped torch.nn.Module scripted and saved from Python nests its parameters in submodule objects
    (c1, l1, ...); named_parameters() - and this reader - report them under dotted names, the names tch's VarStore uses."""
    import torch

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = torch.nn.Conv2d(4, 3, 8, 4)
            self.l1 = torch.nn.Linear(5, 7)
            self.l2 = torch.nn.Linear(7, 2)

        def forward(self, x):
            return x

    torch.manual_seed(3)
    net = Net()
    # a non-contiguous parameter: the reader honours strides
    with torch.no_grad():
        net.l1.weight = torch.nn.Parameter(torch.randn(5, 7).t())
    path = str(tmp_path / "py_export.pt")
    torch.jit.save(torch.jit.script(net), path)
    spec = [(n, tuple(p.shape)) for n, p in net.named_parameters()]
    assert [n for n, _ in spec] == ["c1.weight", "c1.bias", "l1.weight", "l1.bias", "l2.weight", "l2.bias"]
    got = ck.read(path, spec)
    for n, p in net.named_parameters():
        assert np.array_equal(got[n], p.detach().numpy()), n


def test_safetensors_and_archive_are_chosen_by_file_name(ck, tmp_path):
    from safetensors.numpy import load_file
    tensors = {name: fill(i, dims) for i, (name, dims) in enumerate(CNN_SPEC)}
    st, ar = str(tmp_path / "x.safetensors"), str(tmp_path / "x.pt.tch")
    ck.write(st, tensors)
    ck.write(ar, tensors)
    d = load_file(st)
    assert all(np.array_equal(d[k], v) for k, v in tensors.items())
    assert zipfile.is_zipfile(ar) and not zipfile.is_zipfile(st)
    for f in (st, ar):
        back = ck.read(f, CNN_SPEC)
        assert all(np.array_equal(back[k], v) for k, v in tensors.items())


@pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_build/libtorch_archive not built (oracle/build_libtorch_archive.sh)")
def test_cxx_libtorch_loader_agrees(ck, tmp_path):
    """torch::jit::load + named_parameters() from C++, the exact calls behind tch's `VarStore::load`."""
    def fnv(a):
        h = 1469598103934665603
        for b in a.tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return f"{h:016x}"
    tensors = {name: fill(20 + i, dims) for i, (name, dims) in enumerate(CNN_SPEC)}
    path = str(tmp_path / "pi.pt.tch")
    ck.write(path, tensors)
    out = subprocess.run([TOOL, "read", path], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == len(tensors)
    for line, (name, v) in zip(out, tensors.items()):
        parts = line.split()
        assert parts[0] == name and int(parts[1]) == v.ndim and tuple(int(x) for x in parts[2:2 + v.ndim]) == v.shape
        assert parts[-1] == fnv(v), name
    # and the fixture generator's writer against the library's reader, fresh (not the committed file)
    f2 = str(tmp_path / "fresh.pt.tch")
    subprocess.run([TOOL, "write", f2, "a.b", "2,3", "c", "4"], check=True)
    got = ck.read(f2, [("a.b", (2, 3)), ("c", (4,))])
    assert np.array_equal(got["a.b"], fill(0, (2, 3))) and np.array_equal(got["c"], fill(1, (4,)))


def test_malformed_archives_return_errors_instead_of_terminating(ck, tmp_path):
    """The reader is driven by lengths and counts stored in the file.  Truncations, corrupted length fields and wild tensor
    shapes must come back as BDR_ERR_IO through the C ABI - never as an exception crossing `extern "C"` (which would
    terminate the host process) or an out-of-bounds read."""
    import struct
    from border_amd import BdrError
    good = open(os.path.join(GOLD, "libtorch_mlp_qnet.pt.tch"), "rb").read()
    cases = {}
    for cut in (30, 100, len(good) // 2, len(good) - 30, len(good) - 5):
        cases[f"cut{cut}"] = good[:cut]
    eocd = good.rfind(b"PK\x05\x06")
    cd_off = struct.unpack_from("<I", good, eocd + 16)[0]
    b = bytearray(good); struct.pack_into("<H", b, cd_off + 28, 0xFFFF); cases["name_len"] = bytes(b)           # file-name length
    b = bytearray(good); struct.pack_into("<H", b, cd_off + 30, 0xFFFF); cases["extra_len"] = bytes(b)          # extra-field length
    b = bytearray(good); struct.pack_into("<I", b, cd_off + 42, 0xFFFFFFF0); cases["local_off"] = bytes(b)      # local-header offset
    b = bytearray(good); struct.pack_into("<I", b, cd_off + 20, 0x7FFFFFFF); struct.pack_into("<I", b, cd_off + 24, 0x7FFFFFFF)
    cases["entry_size"] = bytes(b)
    z64 = good.rfind(b"PK\x06\x06")       # libtorch writes a zip64 end record; its fields override the 16/32-bit ones
    assert z64 > 0
    b = bytearray(good); struct.pack_into("<Q", b, z64 + 32, 1 << 40); cases["entry_count"] = bytes(b)
    b = bytearray(good); struct.pack_into("<Q", b, z64 + 48, (1 << 63) + 5); cases["cd_offset"] = bytes(b)
    # a tensor whose pickled shape is astronomically large (sizes are BININT / LONG1 operands in data.pkl)
    pk = good.find(b"data.pkl")
    i = good.find(b"(K\x40K\x04t(K\x04K\x01t")      # the (64, 4) size and (4, 1) stride tuples of mlp.ln0.weight
    assert i > 0
    b = bytearray(good); b[i + 2] = 0xFF; b[i + 4] = 0xFF; cases["reads_past_storage"] = bytes(b)
    for name, data in cases.items():
        f = tmp_path / f"{name}.pt.tch"
        f.write_bytes(data)
        with pytest.raises(BdrError) as e:
            ck.read(str(f), MLP_SPEC)
        assert e.value.code == 5, (name, str(e.value))
    # huge declared shape through a hand-made pickle: rebuild the archive with a patched data.pkl
    import io
    zin = zipfile.ZipFile(io.BytesIO(good))
    out = tmp_path / "huge_shape.pt.tch"
    with zipfile.ZipFile(out, "w", zipfile.ZIP_STORED) as zo:
        for info in zin.infolist():
            data = zin.read(info.filename)
            if info.filename.endswith("data.pkl") and "/code/" not in info.filename and ".data/" not in info.filename:
                j = data.find(b"(K\x40K\x04t")
                assert j > 0
                data = data[:j] + b"(J\xff\xff\xff\x7fJ\xff\xff\xff\x7ft" + data[j + 6:]   # (2^31-1, 2^31-1)
            zo.writestr(info.filename, data)
    with pytest.raises(BdrError) as e:
        ck.read(str(out), MLP_SPEC)
    assert e.value.code == 5

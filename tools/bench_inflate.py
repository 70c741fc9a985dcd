"""k_bgzf_inflate alone: the BGZF blocks of a config-3 BAM resident in HBM, timed with HIP events.
    python tools/bench_inflate.py [reads] [level] [iters]"""
import ctypes as C, json, os, struct, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np, torch
from coral_amd import bam, synth, _lib

if os.environ.get('CORAL_LIB'):       # a variant build (ring size / occupancy experiments); before ANYTHING loads the library:
    _lib.LIB_PATH = os.path.abspath(os.environ['CORAL_LIB'])      # _lib.lib() caches the first library it opens
    print("library under test:", _lib.LIB_PATH, file=sys.stderr)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rec = synth.generate(synth.scaled_config("cfg3", n), "cuda:0", chunk_pieces=200000).to("cpu")
d = tempfile.mkdtemp(prefix="coral_infl_")
p = os.path.join(d, "x.bam")
bam.write_bam_native(rec, p, seed=1, level=level)
raw = np.fromfile(p, dtype=np.uint8)
desc, at, out_off = [], 0, 0
while at + 18 <= len(raw):
    xlen = int(raw[at + 10]) | int(raw[at + 11]) << 8
    bsize = (int(raw[at + 16]) | int(raw[at + 17]) << 8) + 1
    isize = int.from_bytes(raw[at + bsize - 4: at + bsize].tobytes(), "little")
    desc.append((at + 12 + xlen, bsize - 12 - xlen - 8, out_off, isize))
    out_off += isize
    at += bsize
L = _lib.lib()
dev = "cuda:0"
comp = torch.from_numpy(np.concatenate([raw, np.zeros(4096, dtype=np.uint8)])).to(dev)
t_desc = torch.tensor(desc, dtype=torch.int64).to(torch.int32).contiguous().to(dev)
out = torch.empty(out_off + 4096, dtype=torch.uint8, device=dev)
status = torch.zeros(len(desc), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ms = []
for it in range(iters + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert L.coral_bgzf_inflate(comp.data_ptr(), t_desc.data_ptr(), len(desc), out.data_ptr(), status.data_ptr(), stream) == 0
    e1.record()
    e1.synchronize()
    if it:
        ms.append(e0.elapsed_time(e1))
ablate = int(os.environ.get('CORAL_INFLATE_ABLATE', '0'))
assert ablate or int(status.abs().sum()) == 0
import zlib
# spot check three blocks against zlib
for k in (() if ablate else (0, len(desc) // 2, len(desc) - 2)):
    so, sl, do, n_out = desc[k]
    assert zlib.decompress(raw[so:so + sl].tobytes(), -15) == out[do:do + n_out].cpu().numpy().tobytes()
best = min(ms)
print(json.dumps({"ablate": ablate, "reads": n, "level": level, "blocks": len(desc), "compressed_MB": len(raw) / 1e6, "inflated_MB": out_off / 1e6,
                  "ms": [round(x, 3) for x in ms], "GB_per_s_out": out_off / best / 1e6, "us_per_block_wave": best * 1e3 / (len(desc) / (256 * 20))}))
import shutil; shutil.rmtree(d, ignore_errors=True)

"""A config-3 BAM for the lab programs:   python tools/lab/make_bam.py out.bam [reads] [zlib level]"""
import sys
sys.path.insert(0, ".")
from coral_amd import bam, synth
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
rec = synth.generate(synth.scaled_config("cfg3", n), "cuda:0", chunk_pieces=200000).to("cpu")
bam.write_bam_native(rec, sys.argv[1], seed=1, level=int(sys.argv[3]) if len(sys.argv) > 3 else 1)

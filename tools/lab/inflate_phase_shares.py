"""Where a wave of k_bgzf_inflate spends a block's time: dynamic header / table build / symbol loop, from s_memrealtime marks in a
TIMING build of the kernel (status[b] carries one of the three sums; not the product library).  Lab tool."""
import os, sys, runpy
os.environ['CORAL_INFLATE_ABLATE'] = '1'
sys.argv = ['bench_inflate.py', '30000', '1', '2']
g = runpy.run_path('tools/bench_inflate.py')
st = g['status'].cpu().numpy().astype('float64')
h, b, c = st[0::3].mean(), st[1::3].mean(), st[2::3].mean()
print("ticks of 10 ns per block (means): header %.0f  table build %.0f  symbol loop %.0f  -> shares %.3f %.3f %.3f" % (h, b, c, h / (h + b + c), b / (h + b + c), c / (h + b + c)))

# launch time of coral_cigar_scan at cfg3 for ring depths x record-group sizes (workgroups per CU = what the occupancy query allows)
for cfg in "6 2" "6 4" "6 8" "6 16" "6 24" "4 8" "8 8" "12 8"; do set -- $cfg; echo "== RING $1 GROUP $2"; CORAL_SCAN_RING=$1 CORAL_SCAN_GROUP=$2 timeout -k 10 120 python tools/microbench_scan.py 2>&1 | grep -E "back to back" | head -2; done

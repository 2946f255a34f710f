cd $GRAFT_REPO_ROOT
L=calamity_amd/csrc/libcalamity_hip.so
for cfg in tutorial hera37; do
  echo "== $cfg mfma"; timeout -k 10 200 python tools/kbench.py --config $cfg --layout shared --cache /tmp/kb_$cfg.pkl --steps 200 $L 2>&1 | cut -c1-200
  echo "== $cfg valu"; CALAMITY_HIP_NO_MFMA=1 timeout -k 10 200 python tools/kbench.py --config $cfg --layout shared --cache /tmp/kb_$cfg.pkl --steps 200 $L 2>&1 | cut -c1-200
done

# usage (GPU box): bash scripts/exp_conv_plans.sh   -- times the two-image UNet pass (host ahead of the device) under
# PXT_CONV_PLAN overrides "layer:cfg:splits;..." (layers 1..16; 10-12 = conv5_x at 30x40, 13-16 = decoder)
cd $GRAFT_REPO_ROOT
run() { printf "%-44s" "[$1]"; PXT_CONV_PLAN="$1" python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
run ""
for c in "1:2" "1:4" "1:8" "4:2" "4:4" "14:2" "14:4" "15:2" "15:4" "11:4"; do
  cfg=${c%%:*}; sp=${c##*:}
  run "10:$cfg:$sp;11:$cfg:$sp;12:$cfg:$sp"
done
for c in "2:4" "2:8" "2:16" "17:4" "17:8" "17:16"; do cfg=${c%%:*}; sp=${c##*:}; run "13:$cfg:$sp"; done
for c in "2:1" "2:3" "2:5" "17:1" "17:2" "17:4"; do cfg=${c%%:*}; sp=${c##*:}; run "14:$cfg:$sp"; done
for c in "2:2" "17:1" "17:2"; do cfg=${c%%:*}; sp=${c##*:}; run "15:$cfg:$sp"; done
run ""

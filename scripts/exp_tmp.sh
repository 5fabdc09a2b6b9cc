#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
run() { printf "%-44s" "[$1]"; PXT_CONV_PLAN="$1" python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
run ""
for c in "22:1" "22:2" "23:1" "23:2" "24:1" "24:2" "24:4" "18:2" "18:4"; do
  cfg=${c%%:*}; sp=${c##*:}
  run "10:$cfg:$sp;11:$cfg:$sp;12:$cfg:$sp"
done
run ""
# 60x80 layers (7 = 256->512 no pool, 8 = 512->512, 9 = 512->512 pooled output)
for c in "22:1" "23:1" "24:1"; do cfg=${c%%:*}; sp=${c##*:}; run "7:$cfg:$sp;8:$cfg:$sp"; done
run ""

#!/usr/bin/env bash
# Identify the GPU box a gpurun call landed on (results have differed between boxes: keep the identity next to every log).
{
  echo "host $(hostname) kernel $(uname -r)"
  cat /sys/class/kfd/kfd/topology/nodes/*/name 2>/dev/null | tr '\n' ' '; echo
  for d in /sys/class/drm/card*/device; do
    [ -f "$d/unique_id" ] && echo "unique_id $(cat $d/unique_id) vbios $(cat $d/vbios_version 2>/dev/null) $(cat $d/current_compute_partition 2>/dev/null) $(cat $d/current_memory_partition 2>/dev/null)"
  done
  /opt/rocm/bin/rocm-smi --showuniqueid --showclocks --showtemp --showpower 2>/dev/null | grep -E "Unique|sclk|mclk|Temperature \(Sensor junction|Power" | head -8
  python - <<'PY' 2>/dev/null
import torch
p = torch.cuda.get_device_properties(0)
print("torch:", p.name, "CUs", p.multi_processor_count, "mem GiB", round(p.total_memory / 2**30, 1), "gcn", getattr(p, "gcnArchName", "?"))
PY
} 2>&1

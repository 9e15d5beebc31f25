#!/bin/bash
# What the GPU box exposes about its topology, clocks and power (read-only probe).
out=gpurun_out/probe_host.txt
{
echo "== nproc / affinity"; nproc; python3 -c "import os;print(len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:64])"
echo "== lscpu"; lscpu | head -30
echo "== numa nodes"; ls /sys/devices/system/node/ | head; for n in /sys/devices/system/node/node*; do echo $n $(cat $n/cpulist); done
echo "== kfd nodes"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "-- $n"; cat $n/gpu_id 2>/dev/null; grep -E "simd_count|location_id|domain|drm_render_minor|unique_id|cpu_cores_count" $n/properties 2>/dev/null; done
echo "== drm"; for d in /sys/class/drm/renderD*; do echo $d $(readlink -f $d/device) numa=$(cat $d/device/numa_node 2>/dev/null) cpus=$(cat $d/device/local_cpulist 2>/dev/null); done
echo "== hwmon"; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $d; ls $d | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input freq2_input temp1_input; do [ -r $d/$f ] && echo $f=$(cat $d/$f); done; done
echo "== pp_dpm_sclk"; for d in /sys/class/drm/card*/device; do echo $d; cat $d/pp_dpm_sclk 2>&1 | head -5; cat $d/current_link_speed $d/current_link_width 2>&1; done
echo "== env"; env | grep -E "HIP|ROCR|CUDA|OMP|GPU_" 
echo "== amdsmi"; python3 -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)" 2>&1 | tail -1
echo "== rocm-smi"; timeout 20 rocm-smi --showtopo 2>&1 | head -40; timeout 20 rocm-smi --showbus --showpower --showclocks 2>&1 | head -40
echo "== torch"; python3 - <<'P'
import torch
print(torch.cuda.device_count())
p=torch.cuda.get_device_properties(0); print(p.name, getattr(p,'pci_bus_id',None), getattr(p,'pci_device_id',None), getattr(p,'pci_domain_id',None), p.multi_processor_count)
P
} > $out 2>&1
echo done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( lscpu | grep -i "numa\|socket\|^CPU(s)\|model name"; cat /sys/fs/cgroup/cpu.max; for d in /sys/bus/pci/devices/*; do c=$(cat $d/class 2>/dev/null); if [ "$c" = "0x038000" ] || [ "$c" = "0x030200" ] || [ "$c" = "0x120000" ]; then echo "$d class $c numa $(cat $d/numa_node)"; fi; done; nproc; taskset -p $$ ) > gpurun_out/r04/numa_topology.txt 2>&1
cat gpurun_out/r04/numa_topology.txt
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -5

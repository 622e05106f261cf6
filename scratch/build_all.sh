#!/bin/bash
# product library + the stamped diagnostic build (absolute paths: safe from any cwd)
set -e
make -j8 -C /root/repo/dp_gp_lvm_amd/csrc > /tmp/make.log 2>&1 || { grep -E " error|Error" /tmp/make.log | head -20; exit 1; }
/root/repo/scratch/build_stamps.sh > /tmp/stamps.log 2>&1 || { grep -E " error" /tmp/stamps.log | head; exit 1; }
ls -la --time-style=+%T /root/repo/dp_gp_lvm_amd/csrc/libdpgp_hip.so /root/repo/scratch/libdpgp_hip_stamps.so | awk '{print $6, $7}'

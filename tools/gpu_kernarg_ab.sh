for v in 0 1; do echo "HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v python tools/latency.py 1 256 2>&1 | grep -v amdgpu.ids; done
for v in 0 1; do echo "HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v python tools/latency.py 1 256 2>&1 | grep -v amdgpu.ids; done

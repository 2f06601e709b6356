set -e
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_msm.py tests/test_gpu_field.py 2>&1 | tail -3
echo "== f29=1"; JFHIP_MSM_F29=1 timeout -k 10 120 python -u scripts/msm_probe.py 24 22 20 2>&1 | tail -3

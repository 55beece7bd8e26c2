GPE_EXTRA_CXXFLAGS="-DGPE_OS_STAMPS" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || echo build failed
timeout -k 10 120 python scripts/time_step.py 1000000 300 2>&1 | grep "os stamps\|^n=" | tail -4 | cut -c1-220
timeout -k 10 120 python scripts/time_step.py 100000000 30 on 2>&1 | grep "os stamps\|^n=" | tail -3 | cut -c1-220
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1

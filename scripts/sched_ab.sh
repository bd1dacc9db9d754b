for r in 1 2; do
echo "base:   $(timeout 200 python scripts/time_k67.py B 2>&1 | tail -1)"
echo "maxilp: $(SPLAT_HIP_LIB=$PWD/splatam_amd/lib_v1/libsplat_hip.so timeout 200 python scripts/time_k67.py B 2>&1 | tail -1)"
echo "memcl:  $(SPLAT_HIP_LIB=$PWD/splatam_amd/lib_v2/libsplat_hip.so timeout 200 python scripts/time_k67.py B 2>&1 | tail -1)"
done

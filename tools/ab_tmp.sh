export MI355RT_ALLOW_DIAGNOSTIC=1
export AB_SIZES=1920x1080,3840x2160
for i in 1 2; do
MI355RT_LIB=tools/bin/libmi355rt_prev.so python tools/ab_flags.py prev=0 2>&1 | grep -v amdgpu.ids
MI355RT_LIB=tools/bin/libmi355rt_box65.so python tools/ab_flags.py box65=0 2>&1 | grep -v amdgpu.ids
MI355RT_LIB=tools/bin/libmi355rt_box9.so python tools/ab_flags.py box9=0 2>&1 | grep -v amdgpu.ids
python tools/ab_flags.py box6=0 2>&1 | grep -v amdgpu.ids
done

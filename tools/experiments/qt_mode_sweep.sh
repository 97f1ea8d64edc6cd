export ORBX_STRICT_ENV=1
for b in 32 64 128 256 512; do for m in 0 3; do echo "B=$b mode=$m"; ORBX_QT_PER_LEVEL=$m timeout -k 10 240 python tools/time_single_frame.py $b | grep device-resident || exit 1; done; done

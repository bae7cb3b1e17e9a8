for v in "" variants/lib_fir_abl1.so variants/lib_fir_abl2.so variants/lib_fir_abl3.so; do
echo "--- lib=$v"; TD_HOTPATH_LIB=$v timeout 120 python tools/time_decode.py 2>&1 | grep "W 1000 hop  100"
done

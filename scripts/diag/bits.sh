# same bits from an experimental build as from the shipped library?  usage: bits.sh name [K,K,...]
cd $GRAFT_REPO_ROOT
export BITS_K=${2:-8}
python3 scripts/diag/bits.py > /tmp/bits_shipped.txt 2>&1
MCHIP_ALLOW_PARTIAL_ABI=1 MCHIP_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/libmulticlust_hip_$1.so python3 scripts/diag/bits.py > /tmp/bits_$1.txt 2>&1
if diff /tmp/bits_shipped.txt /tmp/bits_$1.txt > /tmp/bits.diff; then echo "bits: $1 == shipped on $(wc -l < /tmp/bits_shipped.txt) fits"; else echo "bits: $1 DIFFERS"; sed -n 1,20p /tmp/bits.diff; fi
tail -3 /tmp/bits_$1.txt

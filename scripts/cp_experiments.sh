ALL="-DCP_X_NOA -DCP_X_NOFETCH -DCP_X_NOPATCH -DCP_X_NOBAR"
for v in "$ALL" "$ALL -DCP_X_NOFOLD" "$ALL -DCP_X_NOFOLD -DCP_X_NOSTORE" "-DCP_X_NOFOLD" "-DCP_X_NOSTORE"; do
  bash scripts/cnn_variant.sh "$v" "VPK_ALGORITHM=2 python scripts/time_cnn.py --passes 6 102 2>/dev/null | head -1 | cut -c60-175"
done

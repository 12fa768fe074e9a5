for g in 1 2 3 4 6 8 12; do echo "group $g: $(VPK_CONV1_GROUP=$g python3 scripts/time_cnn.py --passes 8 102 2>/dev/null | head -1 | grep -o "'conv1': [0-9.]*")"; done

for p in 0 1 2; do echo "PROFILE=$p"; PROFILE=$p ONLY=EMPS timeout -k 10 200 python tools/config_times.py 2>&1 | grep EMPS; done
echo chunk4; PROFILE=0 CHUNK=4 ONLY=EMPS timeout -k 10 200 python tools/config_times.py 2>&1 | grep EMPS
echo nooverlap; PROFILE=0 NO_OVERLAP=1 ONLY=EMPS timeout -k 10 200 python tools/config_times.py 2>&1 | grep EMPS
echo vehicle; PROFILE=0 ONLY=Vehicle timeout -k 10 200 python tools/config_times.py 2>&1 | grep Vehicle

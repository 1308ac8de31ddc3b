R=$PWD; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ph_c
rocprofv3 --kernel-trace --stats -d /tmp/ph_c -o p -- python3 $R/profiles/scratch/first_lookup_cold.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/ph_c/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
out = []
warm = False
for i, (name, s, e) in enumerate(rows):
    if 'reduce_kernel' in name and 'sum' in name.lower(): warm = True
    if 'k_frame_init' in name:
        # first grid launch after this init
        for name2, s2, e2 in rows[i + 1:i + 12]:
            if 'k_frame_grid' in name2:
                m = [(e3 - s3) / 1e3 for n3, s3, e3 in rows[i + 1:i + 12] if 'k_frame_march' in n3][:1]
                out.append(("table just read" if warm else "as is", round((e2 - s2) / 1e3, 1), m))
                break
        warm = False
for o in out: print(o)
PY

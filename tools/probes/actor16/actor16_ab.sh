# A/B of the wave-private actor kernel against k_mlp; extra arguments = variant libraries under tools/probes/_variants
cd /root/repo
python tools/probes/actor16_check.py /tmp/ref.pt 2>&1 | grep "actor pass"
for v in "$@"; do
  A16_LIB=/root/repo/tools/probes/_variants/$v.so CM_ACTOR_KERNEL=wave16 python tools/probes/actor16_check.py /tmp/w16.pt /tmp/ref.pt 2>&1 | grep -E "actor pass|max rel" | cut -c1-110
done

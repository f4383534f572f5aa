import glob, os, sys
import pandas as pd
d = sys.argv[1]
kt = pd.read_csv(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0])
_m = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
mc = pd.read_csv(_m[0]) if _m else pd.DataFrame(columns=["Start_Timestamp", "End_Timestamp", "Direction"])
kt["name"] = kt["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("css::", "").str.replace(r"<.*", "", regex=True).str.replace("void ", "")
end = kt["End_Timestamp"].max()
t0 = end - 22_000_000   # last 22 ms
k = kt[kt["Start_Timestamp"] >= t0]; m = mc[mc["Start_Timestamp"] >= t0]
rel = lambda x: (x - t0) / 1e3
print("copies:")
for _, r in m.sort_values("Start_Timestamp").iterrows():
    print(f"  {r['Direction'][12:]:16s} {rel(r['Start_Timestamp']):9.1f} -> {rel(r['End_Timestamp']):9.1f}")
print("markers per queue (features = start of a lane chain, wave_ola = end of a tail):")
for q, g in k.groupby("Queue_Id"):
    g = g.sort_values("Start_Timestamp")
    marks = g[g["name"].isin(["features_kernel", "wave_ola_kernel", "stft_fft_kernel", "beamform_kernel"])]
    print(f" queue {q}: {len(g)} kernels;", " ".join(f"{r['name'][:4]}@{rel(r['Start_Timestamp']):.0f}" for _, r in marks.iterrows()))
# everything between the end of one pass's lanes and the start of the next pass's
b = k[k["name"] == "beamform_kernel"].sort_values("Start_Timestamp")
s2 = k[k["name"] == "stft_fft_kernel"].sort_values("Start_Timestamp")
if len(b) > 3 and len(s2) > 3:
    t_a = b.iloc[2]["End_Timestamp"] - 50_000
    nxt = s2[s2["Start_Timestamp"] > t_a]
    t_b = nxt.iloc[min(2, len(nxt) - 1)]["End_Timestamp"] + 100_000
    w = kt[(kt["Start_Timestamp"] >= t_a) & (kt["Start_Timestamp"] <= t_b)].sort_values("Start_Timestamp")
    print(f"kernels between {rel(t_a):.0f} and {rel(t_b):.0f} us:")
    for _, r in w.iterrows():
        print(f"  q{r['Queue_Id']} {rel(r['Start_Timestamp']):8.1f} -> {rel(r['End_Timestamp']):8.1f}  {r['name']}")

import json,sys
d=json.load(open(sys.argv[1]))
print("value %.3f M  ms/step %.4f  frac %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"]))
print("sustained", d.get("sustained",{}).get("frames_per_s"))
w=d["other_workloads"]["wide_slices_1500k"]
for k,v in w["pictures_per_call"].items():
    if k!="what": print("wide pics/call",k, "%.2f M"%(v["frames_per_s"]/1e6), v.get("serial_stage_ms"))
print("vmedia", d["other_workloads"].get("vmedia_x1024",{}).get("frames_per_s"))
print("ingest staged", d["ingest"]["pcie_inclusive_frames_per_s"], "in place", d["ingest"].get("in_place_from_page_locked_arena",{}).get("pcie_inclusive_frames_per_s"))
vo=d.get("video_out") or {}
for k,v in vo.items():
    if isinstance(v,dict): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,str))})

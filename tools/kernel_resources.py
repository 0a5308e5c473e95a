import subprocess,sys,re,os
f=sys.argv[1]
# extract the device code object from a host object: section .hip_fatbin holds a clang offload bundle
out=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objcopy','--dump-section','.hip_fatbin=/tmp/rs/fat.bin',f],capture_output=True,text=True)
lst=subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler','--list','--type=o','--input=/tmp/rs/fat.bin'],capture_output=True,text=True).stdout.split()
tgt=[t for t in lst if 'gfx950' in t][0]
subprocess.check_call(['/opt/rocm/lib/llvm/bin/clang-offload-bundler','--unbundle','--type=o','--input=/tmp/rs/fat.bin','--targets='+tgt,'--output=/tmp/rs/dev.co'])
notes=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf','--notes','/tmp/rs/dev.co'],capture_output=True,text=True).stdout
cur={}
rows=[]
for line in notes.split('\n'):
    m=re.match(r'\s+\.(\w+):\s+(.*)',line) or re.match(r'\s+- \.(\w+):\s+(.*)',line)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=='name' and v.startswith('_Z'): cur['name']=v
    if k in ('vgpr_count','sgpr_count','private_segment_fixed_size','group_segment_fixed_size','agpr_count'): cur[k]=v
    if k=='wavefront_size':
        rows.append(cur); cur={}
for r in rows:
    if 'name' not in r: continue
    dn=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    dn=re.sub(r'\(.*','',dn)
    print(f"{dn:60s} vgpr {r.get('vgpr_count'):>4} agpr {r.get('agpr_count','0'):>3} sgpr {r.get('sgpr_count'):>4} scratch {r.get('private_segment_fixed_size'):>5} lds {r.get('group_segment_fixed_size'):>7}")

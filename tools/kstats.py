import re,sys
for l in open(sys.argv[1]):
    if re.search(r'schur|gj_|sym_gemv|dense_gemm',l):
        parts=l.strip().split(',')
        m=re.search(r'(k_\w+(<[^>]*>)?)',parts[0]); name=m.group(1) if m else parts[0][:60]
        print("%-44s calls %4s avg %9s min %7s"%(name[:44],parts[1],parts[3],parts[4]))

# generates the GCC inline-asm body of a 6x64 Montgomery product (mulx/adcx/adox, no-carry variant: top bit of p is free)
T=['%[t0]','%[t1]','%[t2]','%[t3]','%[t4]','%[t5]']
A='%[A]'; AX='%[ax]'; BX='%[bx]'
L=[]
def e(s): L.append(s)
for i in range(6):
    e('xorl %k[ax], %k[ax]')            # clears CF and OF
    e(f'movq {8*i}(%[y]), %%rdx')
    if i==0:
        e(f'mulxq 0(%[x]), {T[0]}, {T[1]}')
        for j in range(1,6):
            hi = T[j+1] if j<5 else A
            e(f'mulxq {8*j}(%[x]), {AX}, {hi}')
            e(f'adoxq {AX}, {T[j]}')
        e(f'movl $0, %k[ax]')
        e(f'adoxq {AX}, {A}')
    else:
        e(f'mulxq 0(%[x]), {AX}, {A}')
        e(f'adoxq {AX}, {T[0]}')
        for j in range(1,6):
            e(f'adcxq {A}, {T[j]}')
            e(f'mulxq {8*j}(%[x]), {AX}, {A}')
            e(f'adoxq {AX}, {T[j]}')
        e(f'movl $0, %k[ax]')
        e(f'adcxq {AX}, {A}')
        e(f'adoxq {AX}, {A}')
    # reduction
    e(f'movq %[ninv], %%rdx')
    e(f'imulq {T[0]}, %%rdx')
    e('xorl %k[ax], %k[ax]')
    e(f'mulxq 0(%[p]), {AX}, {BX}')
    e(f'adcxq {T[0]}, {AX}')
    e(f'movq {BX}, {T[0]}')
    for j in range(1,6):
        e(f'adcxq {T[j]}, {T[j-1]}')
        e(f'mulxq {8*j}(%[p]), {AX}, {T[j]}')
        e(f'adoxq {AX}, {T[j-1]}')
    e(f'movl $0, %k[ax]')
    e(f'adcxq {AX}, {T[5]}')
    e(f'adoxq {A}, {T[5]}')
print('\n'.join('      "%s\\n\\t"' % s for s in L))

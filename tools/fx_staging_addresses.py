"""tools/fx_staging_addresses.py -- host-side check of every staging load address of fx_blur_u8 (strip / image resources, fx_strip_range,
the clamp of lanes past the window) over a list of shapes and window sizes: nothing may fall outside the frame or the written part of a strip"""
import itertools
def left_strips(pada): return (pada+127)//128 if pada>0 else 1
def right_strips(cols,pada):
    chunks=(cols+127)//128; n=0; xc=chunks-1
    while xc>=left_strips(pada) and 128*xc+128+pada>cols: n+=1; xc-=1
    return n
def strip_range(xc,cols,pada,per):
    x0=128*xc; left=x0-pada<0; right=x0+128+pada>cols
    if left and right: return 0,per
    if left:
        gl=(pada-x0+3)//4; h=(gl+7)//8; return 0,min(h,per)
    if right:
        gb=(cols-(x0-pada))//4; return gb//8,per
    return 0,0
bad=0
for nkb in (3,4,5,6,7,8,9,10,11):
    pada=8*(nkb-2); win=128+2*pada; gpr=win//4; per=(gpr+7)//8
    for rows,cols in [(70,68),(100,100),(67,256),(300,72),(66,132),(270,480),(131,152),(97,644),(200,332),(150,260),(2160,3840),(1080,1921),(90,4004),(333,251),(100,300),(100,128),(160,100),(80,200),(256,384),(332,516),(400,421),(64,129),(64,127),(64,200),(64,199),(64,201),(50,273),(50,400)]:
        pad=min(pada, min(rows,cols)-1)
        chunks=(cols+127)//128; nleft=left_strips(pada); nright=right_strips(cols,pada)
        fb=rows*cols*3
        for xc in range(chunks):
            sidx = xc if xc<nleft else (nleft+xc-(chunks-nright) if xc>=chunks-nright else -1)
            lo,hi=(strip_range(xc,cols,pada,per) if sidx>=0 else (0,0))
            x0=128*xc
            for k in range(per):
                fs = lo<=k<hi
                for g0 in range(8):
                    inn = (gpr%8==0) or k<per-1 or g0<gpr%8
                    kk = k if inn else (k-1 if k>0 else 0)
                    for r in (0,rows-1):
                        if fs:
                            off=r*3*win+12*g0+96*kk
                            if off+12>rows*3*win: print("strip oob",nkb,rows,cols,xc,k,g0,r); bad+=1
                            # group must be written: within [8lo, min(8hi,gpr)) or clamped lanes
                            gi=g0+8*kk
                            if inn and not (8*lo<=gi<min(8*hi,gpr)): print("strip unwritten",nkb,rows,cols,xc,k,g0); bad+=1
                        else:
                            a=3*(x0-pada)+r*3*cols+12*g0+96*kk   # relative to frame start
                            size=fb-3*(x0-pada); offs=r*3*cols+12*g0+96*kk
                            if offs+12<=size:   # passes bounds check -> must be inside the frame
                                if a<0 or a+12>fb: print("image oob",nkb,rows,cols,xc,k,g0,r,a); bad+=1
                            if inn:
                                # pixels must be inside [0, cols)
                                px=x0-pada+4*(g0+8*k)
                                if px<0 or px+3>=cols:
                                    # reading beyond row end is only ok if those positions meet zero taps: position index >= 128+2*pad_real...
                                    if px<0: print("image mirrored-left needed",nkb,rows,cols,xc,k,g0); bad+=1
print("bad",bad)

"""TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates the reference's netlib ingestion: tools/preprocessing.sh:2-9 runs
`emps` (decompress) and then tools/mps_load.py, which needs `pulp` (absent in
this image).  This module restates what mps_load.py:51-89 writes for a
fixed-format MPS file read by pulp's `LpProblem.fromMPS` (pulp 2.x:
whitespace-split fields, variables in order of first appearance in COLUMNS,
default lower bound 0, `variables()` sorted by name).  The result is pinned
by replaying the reference's netlib goldens (oracle/pin_oracle.py).
"""


def _f(x):  # mps_load.py:14-15
    return format(x, "f")


def read_mps(path):
    rows = []          # (sense, name) in ROWS order, objective excluded
    row_coef = {}      # name -> list[(var, value)]
    obj_name = None
    obj_coef = []
    obj_const = 0.0
    rhs = {}
    var_order = []
    bounds = {}        # var -> [low, up]
    mode = None
    with open(path) as f:
        for raw in f:
            if not raw.strip() or raw.startswith("*"):
                continue
            parts = raw.split()
            if not raw[0].isspace():
                mode = parts[0]
                if mode == "ENDATA":
                    break
                continue
            if mode == "ROWS":
                sense, name = parts[0], parts[1]
                if sense == "N":
                    if obj_name is None:
                        obj_name = name
                else:
                    rows.append((sense, name))
                    row_coef[name] = []
            elif mode == "COLUMNS":
                if len(parts) >= 3 and parts[1] == "'MARKER'":
                    continue
                var = parts[0]
                if var not in bounds:
                    bounds[var] = [0.0, None]
                    var_order.append(var)
                for k in range(1, len(parts) - 1, 2):
                    rname, val = parts[k], float(parts[k + 1])
                    if rname == obj_name:
                        obj_coef.append((var, val))
                    elif rname in row_coef:
                        row_coef[rname].append((var, val))
            elif mode == "RHS":
                if len(parts) % 2 == 1:
                    parts = parts[1:]
                for k in range(0, len(parts) - 1, 2):
                    rname, val = parts[k], float(parts[k + 1])
                    if rname == obj_name:
                        obj_const = -val
                    else:
                        rhs[rname] = val
            elif mode == "BOUNDS":
                kind, var = parts[0], parts[2]
                if kind == "FR":
                    bounds[var] = [None, None]
                elif kind == "BV":
                    bounds[var] = [0.0, 1.0]
                else:
                    val = float(parts[3])
                    if kind == "LO":
                        bounds[var][0] = val
                    elif kind == "UP":
                        bounds[var][1] = val
                    elif kind == "FX":
                        bounds[var] = [val, val]
                    elif kind == "MI":
                        bounds[var][0] = None
                    elif kind == "PL":
                        bounds[var][1] = None
                    else:
                        raise ValueError("unsupported bound type " + kind)
            elif mode == "RANGES":
                raise ValueError("RANGES section not supported (none of the shipped netlib files has one)")
    return dict(rows=rows, row_coef=row_coef, obj_coef=obj_coef, obj_const=obj_const, rhs=rhs,
                var_order=var_order, bounds=bounds)


def _expr(coefs, const, uni):  # mps_load.py:18-34
    s = ""
    for k, (var, val) in enumerate(coefs):
        if val < 0:
            s += _f(val) + "*" + uni[var]
        else:
            s += ("+" if k > 0 else "") + _f(val) + "*" + uni[var]
    s += ("+" if coefs else "") + _f(const)
    return s


def mps_to_text(path):
    p = read_mps(path)
    uni = {v: "x%d" % k for k, v in enumerate(p["var_order"])}   # mps_load.py:7-11
    out = ["min " + _expr(p["obj_coef"], p["obj_const"], uni), "st"]
    sense = {"E": "=", "G": ">=", "L": "<="}
    for s, name in p["rows"]:                                     # mps_load.py:37-47
        out.append(_expr(p["row_coef"][name], 0.0, uni) + sense[s] + _f(p["rhs"].get(name, 0.0)))
    for var in sorted(p["var_order"]):                            # mps_load.py:74-89
        lo, up = p["bounds"][var]
        if lo is not None:
            out.append(uni[var] + ">=" + _f(lo))
        if up is not None:
            out.append(uni[var] + "<=" + _f(up))
    return "\n".join(out) + "\n"

"""rust/sublinear_hip.rs (the reference-side binding; source only — no rustc in the image) against include/sublinear_hip.h:
every `extern "C"` declaration and every `#[repr(C)]` struct of the shim must agree with the header it binds — function names, argument
counts, pointer / scalar kinds and widths, constness of pointees, return types; struct field names, order and widths.  The interfaces the
shim implements are `trait SolverAlgorithm` / `SolverState` (src/solver/mod.rs:223-351) over `&dyn Matrix` (src/matrix/mod.rs:25-104)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "sublinear_hip.h"
SHIM = ROOT / "rust" / "sublinear_hip.rs"

C_SCALARS = {"uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "int": "i32", "double": "f64", "float": "f32", "char": "c_char",
             "void": "void", "uint16_t": "u16", "uint8_t": "u8", "int64_t": "i64", "size_t": "usize"}
RUST_SCALARS = {"u64": "u64", "u32": "u32", "i32": "i32", "c_int": "i32", "f64": "f64", "f32": "f32", "c_char": "c_char", "c_void": "void",
                "u16": "u16", "u8": "u8", "i64": "i64", "usize": "usize"}


def _snake(name):        # SlNeumannOptions -> sl_neumann_options, SlCommInfoT -> sl_comm_info_t
    return re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()


def parse_header(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    enums = set(re.findall(r"typedef\s+enum\s*\w*\s*\{[^}]*\}\s*(\w+)\s*;", text))
    opaque = set(m for m in re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{([^}]*)\}\s*(\w+)\s*;", text):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            cm = re.match(r"(const\s+)?(\w+)\s+(.*)$", decl)
            base, rest = cm.group(2), cm.group(3)
            for piece in rest.split(","):
                piece = piece.strip()
                stars = piece.count("*")
                fields.append((piece.replace("*", "").strip(), ctype(("const " if cm.group(1) else "") + base + " " + "*" * stars, enums)))
        structs[m.group(2)] = fields
    body = re.sub(r"typedef\s+(struct|enum)\s*\w*\s*\{[^}]*\}\s*\w+\s*;", "", text)
    funcs = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(sl_\w+)\s*\(([^()]*)\)\s*;", body):
        ret = m.group(1).strip()
        args = [a.strip() for a in m.group(3).split(",")] if m.group(3).strip() not in ("", "void") else []
        parsed = []
        for a in args:
            am = re.match(r"(.*?)(\w+)$", a)                      # type, then the parameter's name
            parsed.append(ctype(am.group(1).strip(), enums))
        funcs[m.group(2)] = (ctype(ret, enums), parsed)
    return structs, funcs, opaque, enums


def ctype(t, enums):
    t = t.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w not in ("const", "struct", "extern")][-1]
    base = "i32" if base in enums else C_SCALARS.get(base, base)          # C enums are ints
    if stars == 0:
        return base
    out = base
    for level in range(stars):
        out = ("*const " if (const and level == 0) else "*mut ") + out     # constness of the pointee matters at the first level only
    return out


def rtype(t):
    t = t.strip()
    m = re.match(r"\*(const|mut)\s+(.*)$", t)
    if m:
        inner = rtype(m.group(2))
        return f"*{m.group(1)} {inner}"
    return RUST_SCALARS.get(t, _snake(t))


def parse_shim(text):
    text = re.sub(r"//.*$", "", text, flags=re.M)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+)\s*\{([^}]*)\}", text):
        fields = []
        for f in m.group(2).split(","):
            f = f.strip()
            if f:
                name, ty = f.split(":", 1)
                fields.append((name.replace("pub", "").strip(), ty.strip()))
        structs[m.group(1)] = fields
    funcs = {}
    for blk in re.findall(r'extern "C" \{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"fn (sl_\w+)\s*\(([^()]*)\)\s*(?:->\s*([^;]+))?;", blk):
            args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
            funcs[m.group(1)] = (rtype(m.group(3)) if m.group(3) else "void", [rtype(a) for a in args])
    return structs, funcs


def compare(header_text, shim_text):
    cs, cf, opaque, _ = parse_header(header_text)
    rs, rf = parse_shim(shim_text)
    problems = []
    for name, fields in rs.items():
        cname = _snake(name)
        if fields == [("_private", "[u8; 0]")]:
            if cname not in opaque:
                problems.append(f"{name}: opaque in the shim, but the header has no opaque `{cname}`")
            continue
        if cname not in cs:
            problems.append(f"{name}: no struct `{cname}` in the header")
            continue
        want = cs[cname]
        got = [(n, rtype(t)) for n, t in fields]
        if got != want:
            problems.append(f"{name} vs {cname}: fields differ\n   shim   {got}\n   header {want}")
    for name, (ret, args) in rf.items():
        if name not in cf:
            problems.append(f"{name}: declared in the shim, not in the header")
            continue
        cret, cargs = cf[name]
        if ret != cret:
            problems.append(f"{name}: return type {ret} vs header {cret}")
        if len(args) != len(cargs):
            problems.append(f"{name}: {len(args)} arguments vs {len(cargs)} in the header")
        elif args != cargs:
            problems.append(f"{name}: argument types\n   shim   {args}\n   header {cargs}")
    return problems, rs, rf


def test_shim_declarations_agree_with_the_header():
    problems, rs, rf = compare(HEADER.read_text(), SHIM.read_text())
    assert not problems, "\n".join(problems)
    assert len(rf) >= 35 and len(rs) >= 10, (len(rf), len(rs))
    for must in ("sl_neumann_solve", "sl_neumann_state_create", "sl_neumann_state_update_rhs", "sl_query_session_estimate", "sl_push_graph_create",
                 "sl_forward_push_acl_with_target", "sl_backward_push_acl_with_source", "sl_neumann_state_create_partitioned", "sl_comm_create"):
        assert must in rf, must


def test_a_swapped_field_or_a_changed_argument_is_noticed():
    """the check must be able to fail: swap two fields of a struct, change a pointer's constness, drop an argument"""
    shim = SHIM.read_text()
    swapped = shim.replace("tolerance: f64, max_iterations: u64, max_terms: u64, series_tolerance: f64,",
                           "tolerance: f64, max_terms: u64, max_iterations: u64, series_tolerance: f64,")
    assert swapped != shim and any("SlNeumannOptions" in p for p in compare(HEADER.read_text(), swapped)[0])
    narrowed = shim.replace("pub struct SlAclResult { push_count: u64, nodes_visited: u64,", "pub struct SlAclResult { push_count: u32, nodes_visited: u64,")
    assert narrowed != shim and any("SlAclResult" in p for p in compare(HEADER.read_text(), narrowed)[0])
    const = shim.replace("fn sl_neumann_state_solution(s: *const SlNeumannState,", "fn sl_neumann_state_solution(s: *mut SlNeumannState,")
    assert const != shim and any("sl_neumann_state_solution" in p for p in compare(HEADER.read_text(), const)[0])
    dropped = shim.replace("fn sl_neumann_state_reset(s: *mut SlNeumannState) -> c_int;", "fn sl_neumann_state_reset() -> c_int;")
    assert dropped != shim and any("sl_neumann_state_reset" in p for p in compare(HEADER.read_text(), dropped)[0])


def test_integration_md_points_at_the_shim_and_quotes_it_verbatim():
    doc = (ROOT / "INTEGRATION.md").read_text()
    assert "rust/sublinear_hip.rs" in doc
    norm = lambda s: re.sub(r"\s+", " ", s).strip()
    shim = norm(SHIM.read_text())
    for block in re.findall(r"```rust\n(.*?)```", doc, flags=re.S):
        for line in block.splitlines():
            if line.strip() and not line.strip().startswith("//"):
                assert norm(line) in shim, f"INTEGRATION.md quotes a line that rust/sublinear_hip.rs does not hold: {line.strip()[:120]}"

"""Host-side mirror of the reference's class surface: argument validation,
initialisation / shuffle parity, sharding helpers, metrics, co-occurrence builder.  CPU only."""
import os

import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import golden
from cymf_amd import BPR, WMF, GloVe, RelMF, _host, dist, metrics, synthetic
from cymf_amd.glove import read_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constructor_defaults_match_reference():
    m = BPR()
    assert (m.num_components, m.learning_rate, m.optimizer, m.weight_decay) == (20, 0.001, "adam", 0.01)
    assert m.W is None and m.H is None
    w = WMF()
    assert (w.num_components, w.weight_decay, w.weight) == (20, 0.01, 10.0)
    g = GloVe()
    assert (g.num_components, g.learning_rate, g.alpha, g.x_max) == (50, 0.01, 0.75, 10.0)
    r = RelMF()
    assert (r.num_components, r.clip_value, r.learning_rate, r.optimizer, r.weight_decay) == (20, 0.1, 0.001, "adam", 0.01)


def test_invalid_optimizer_raises_like_reference():
    with pytest.raises(Exception, match="rmsprop is invalid."):
        BPR(optimizer="rmsprop")
    with pytest.raises(Exception, match="x is invalid."):
        RelMF(optimizer="x")


def test_fit_argument_errors():
    X = sparse.csr_matrix(np.eye(3))
    for cls in (BPR, WMF):
        with pytest.raises(ValueError):
            cls().fit(None)
        with pytest.raises(ValueError):
            cls().fit([[1, 0], [0, 1]])
        with pytest.raises(ValueError):
            cls().fit(X, early_stopping=True)            # no evaluator
    with pytest.raises(ValueError):
        RelMF().fit(None)
    with pytest.raises(ValueError):
        GloVe().fit(None, 1, 1)
    with pytest.raises(TypeError):
        GloVe().fit(np.eye(3), 1, 1)                      # dense is rejected (glove.pyx:88-89)


def test_init_and_shuffle_match_reference_semantics():
    class M:
        W = None
        H = None
    m = M()
    _host.init_factors(m, 7, 9, 4)
    users, pos = _host.reference_shuffle(np.arange(20), np.arange(20) * 2)
    W, H = oracle.reference_init(7, 9, 4)
    u2, p2 = oracle.reference_shuffle(np.arange(20), np.arange(20) * 2)
    assert np.array_equal(m.W, W) and np.array_equal(m.H, H)
    assert np.array_equal(users, u2) and np.array_equal(pos, p2)
    # equals sklearn.utils.shuffle on the same global state (cymf/bpr.pyx:104)
    sk = pytest.importorskip("sklearn.utils")
    np.random.seed(99)
    a, b = sk.shuffle(np.arange(50), np.arange(50) + 100)
    np.random.seed(99)
    c, d = _host.reference_shuffle(np.arange(50), np.arange(50) + 100)
    assert np.array_equal(a, c) and np.array_equal(b, d)


def test_preset_H_only_draws_W_after_seed():
    class M:
        W = None
        H = np.ones((5, 3))
    m = M()
    _host.init_factors(m, 4, 5, 3)
    np.random.seed(4321)
    W = np.random.uniform(-0.1, 0.1, (4, 3)) / 3
    assert np.array_equal(m.W, W) and np.array_equal(m.H, np.ones((5, 3)))


def test_membership_pattern_drops_zeros_and_sorts():
    X = sparse.csr_matrix((np.array([1.0, 0.0, 2.0, 1.0]), np.array([3, 1, 0, 2]), np.array([0, 3, 4])), shape=(2, 4))
    indptr, indices = _host.membership_pattern(X)
    assert indptr.tolist() == [0, 2, 3] and indices.tolist() == [0, 3, 2]


def test_pick_mode():
    assert _host.pick_mode(None, 1) == "exact"
    assert _host.pick_mode(None, 8) == "throughput" and _host.pick_mode(None, 0) == "throughput"
    assert _host.pick_mode("exact", 8) == "exact"
    with pytest.raises(ValueError):
        _host.pick_mode("fast", 1)


def test_user_shards_cover_and_balance():
    X = synthetic.implicit_matrix(5000, 800, 60000, 1)
    for world in (1, 2, 3, 8):
        sh = dist.user_shards(X.indptr, world)
        assert sh[0][0] == 0 and sh[-1][1] == 5000
        assert all(sh[r][1] == sh[r + 1][0] for r in range(world - 1))
        loads = [X.indptr[hi] - X.indptr[lo] for lo, hi in sh]
        assert sum(loads) == X.nnz
        assert max(loads) <= X.nnz / world + X.getnnz(axis=1).max()


def test_metrics_match_reference_fixture_and_oracle():
    g = golden("metrics")
    for r, y in enumerate(g["y"]):
        for c, k in enumerate(g["ks"]):
            k = int(k)
            assert metrics.dcg_at_k(y, k) == pytest.approx(g["dcg"][r, c], rel=1e-14, abs=0)
            assert metrics.recall_at_k(y, k) == pytest.approx(g["recall"][r, c], rel=1e-14, abs=0)
            assert metrics.average_precision_at_k(y, k) == pytest.approx(g["ap"][r, c], rel=1e-14, abs=0)
    # IPS variants reduce to the plain ones at unit propensity
    rs = np.random.RandomState(0)
    y = (rs.rand(40) < 0.2).astype(np.int32)
    p = np.ones(40)
    assert metrics.dcg_at_k_with_ips(y, p, 5) == pytest.approx(oracle.dcg_at_k(y, 5))
    assert metrics.recall_at_k_with_ips(y, p, 5) == pytest.approx(oracle.recall_at_k(y, 5))
    assert metrics.average_precision_at_k_with_ips(y, p, 5) == pytest.approx(oracle.ap_at_k(y, 5))


def test_read_text_matches_reference_loop(tmp_path):
    rs = np.random.RandomState(1)
    vocab = [f"w{i}" for i in range(30)]
    words = [vocab[min(int(abs(rs.normal()) * 8), 29)] for _ in range(400)]
    f = tmp_path / "corpus.txt"
    f.write_text(" ".join(words))
    M, i2w = read_text(str(f), min_count=3, window_size=4)
    # straight restatement of cymf/glove.pyx:199-241 (single line of text, as text8 is)
    from collections import Counter
    cnt = Counter(words)
    w2i, ids = {}, []
    for w in words:
        if cnt[w] >= 3:
            w2i.setdefault(w, len(w2i))
            ids.append(w2i[w])
    V = len(w2i)
    D = np.zeros((V, V))
    for j in range(len(ids)):
        for k in range(max(0, j - 4), j):
            D[ids[j], ids[k]] += 1.0 / abs(j - k)
    assert M.shape == (V, V) and np.allclose(M.toarray(), D, rtol=1e-13, atol=0)
    assert [i2w[i] for i in range(V)] == list(w2i)


def test_read_text_three_lines_counts_are_glued_at_line_ends(tmp_path):
    """cymf/glove.pyx:199-203: counts come from raw.replace("\\n", "<eos>").split(" "), windows from raw.split("\\n").
    Here 'c' occurs three times, but twice glued to a neighbour ("c<eos>b", "b<eos>c") -> count 1 -> dropped at
    min_count=2, and 'b' counts 3 of its 5 occurrences.  A per-line count would keep 'c'."""
    f = tmp_path / "three.txt"
    f.write_text("a b a b c\nb a c a b\nc a b a")
    M, i2w = read_text(str(f), min_count=2, window_size=2)
    assert i2w == {0: "a", 1: "b"}
    # kept ids per line: [0,1,0,1], [1,0,0,1], [0,1,0]; X[cur, prev] += 1/(j-k) for k in (j-2, j-1)
    D = np.zeros((2, 2))
    for ids in ([0, 1, 0, 1], [1, 0, 0, 1], [0, 1, 0]):
        for j in range(len(ids)):
            for k in range(max(0, j - 2), j):
                D[ids[j], ids[k]] += 1.0 / (j - k)
    # by hand: a<-a: .5 (l1) + 1 (l2) + .5 (l3); a<-b: 1 (l1) + 1+.5 (l2) + 1 (l3); b<-a: 1+1 (l1) + 1+.5 (l2) + 1 (l3); b<-b: .5 (l1)
    assert np.array_equal(D, np.array([[2.0, 3.5], [4.5, 0.5]]))
    assert M.shape == (2, 2) and np.array_equal(M.toarray(), D)
    # with min_count=1 all three words are kept, in order of first appearance
    M1, i2w1 = read_text(str(f), min_count=1, window_size=2)
    assert i2w1 == {0: "a", 1: "b", 2: "c"} and M1.shape == (3, 3)
    assert M1[2, 1] == 1.0 + 0.5 and M1[2, 0] == 0.5 + 1.0   # 'c' after "a b" (l1) and after "b a" (l2); line 3 starts with it


def test_read_text_word_only_at_a_line_boundary_raises_keyerror_like_the_reference(tmp_path):
    """`count` is a dict (cymf/glove.pyx:203): 'y' only exists inside the glued token "y<eos>z" -> count['y'] raises."""
    f = tmp_path / "glued.txt"
    f.write_text("x y\nz")
    with pytest.raises(KeyError):
        read_text(str(f), min_count=1, window_size=2)
    g = tmp_path / "trailing_newline.txt"   # the empty last line's "" token is not in count either
    g.write_text("x x x\n")
    with pytest.raises(KeyError):
        read_text(str(g), min_count=1, window_size=2)


def test_save_word2vec_format_bytes(tmp_path):
    """cymf/glove.pyx:164-177: header "V K", then "<word> <str(float64)> ..." per row, '\\n' line ends."""
    from cymf_amd import GloVe
    g = GloVe(num_components=4)
    g.W = np.array([[0.5, -1.25, 3.0, 1e-05], [0.1, 2.0, -0.0, 1e16], [1.0 / 3.0, 7.0, 8.5, -2.5e-300]])
    path = tmp_path / "vec.txt"
    g.save_word2vec_format(str(path), {0: "the", 1: "of", 2: "<eos>"})
    assert path.read_bytes() == (b"3 4\n"
                                 b"the 0.5 -1.25 3.0 1e-05\n"
                                 b"of 0.1 2.0 -0.0 1e+16\n"
                                 b"<eos> 0.3333333333333333 7.0 8.5 -2.5e-300\n")
    # a list works as the index->word map too (the reference only indexes it)
    g.save_word2vec_format(str(path), ["a", "b", "c"])
    assert path.read_text().splitlines()[1].split(" ")[0] == "a"


def test_synthetic_configs_have_the_stated_shape():
    X, K = synthetic.config_matrix("C1")
    assert X.shape == (943, 1682) and X.nnz == 44853 and K == 20
    assert X.has_sorted_indices and (X.data == 1.0).all()
    X2, _ = synthetic.config_matrix("C1")
    assert (X != X2).nnz == 0   # deterministic


def test_movielens_local_loader(tmp_path):
    pd = pytest.importorskip("pandas")
    sk = pytest.importorskip("sklearn.model_selection")
    from cymf_amd.dataset import MovieLens
    with pytest.raises(ValueError):
        MovieLens("ml-10b", root=tmp_path)                       # tests/test_dataset.py:18-20 of the reference
    with pytest.raises(FileNotFoundError):
        MovieLens("ml-100k", root=tmp_path)                      # never downloads
    rs = np.random.RandomState(0)
    n = 3000
    users, items = rs.randint(1, 120, n) * 3, rs.randint(1, 200, n) * 2 + 1      # sparse ids -> reset_id matters
    df = pd.DataFrame({"user": users, "item": items, "rating": rs.randint(1, 6, n), "timestamp": rs.randint(0, 10**9, n)})
    df = df.drop_duplicates(["user", "item"])
    (tmp_path / "ml-100k").mkdir()
    df.to_csv(tmp_path / "ml-100k" / "u.data", sep="\t", header=False, index=False)
    ds = MovieLens("ml-100k", root=tmp_path)
    assert ds.train.shape == ds.valid.shape == ds.test.shape == (ds.num_user, ds.num_item)   # tests/test_dataset.py:15-16
    assert ds.num_user == df.user.nunique() and ds.num_item == df.item.nunique()
    pos = int((df.rating >= 4).sum())
    assert ds.train_size + ds.valid_size + ds.test_size == pos
    assert ds.test_size == int(np.ceil(0.1 * pos)) and ds.valid_size == int(np.ceil(0.1 * (pos - ds.test_size)))
    # literal restatement of cymf/dataset/movielens.py:53-67 on the same file
    ref = pd.read_csv(tmp_path / "ml-100k" / "u.data", sep="\t", names=("user", "item", "rating", "timestamp"))
    for col in ("item", "user"):
        m = {}
        for x in set(ref[col]):
            m.setdefault(x, len(m))
        ref[col] = ref[col].map(lambda x: m[x])
    ref = ref[ref["rating"] >= 4.0].copy()
    ref["rating"] = 1.0
    tr, te = sk.train_test_split(ref, test_size=0.1, random_state=12345)
    tr, va = sk.train_test_split(tr, test_size=0.1, random_state=12345)
    want = sparse.lil_matrix((ds.num_user, ds.num_item))
    for u, i, r in zip(tr["user"].values, tr["item"].values, tr["rating"].values):
        want[u, i] = r
    assert (ds.train.tocsr() != want.tocsr()).nnz == 0
    assert sorted(zip(te.user, te.item)) == sorted(zip(*ds.test.nonzero()))


def test_yahoomusic_local_loader(tmp_path):
    pd = pytest.importorskip("pandas")
    sk = pytest.importorskip("sklearn.model_selection")
    from cymf_amd.dataset import YahooMusic
    with pytest.raises(FileNotFoundError):
        YahooMusic(root=tmp_path)                                # cymf/dataset/yahoomusic.py:23-26 exits; this one raises
    rs = np.random.RandomState(1)
    d = tmp_path / "yahoomusic"
    d.mkdir()
    frames = {}
    for name, n in ((YahooMusic.TRAIN_FILE, 4000), (YahooMusic.TEST_FILE, 600)):
        df = pd.DataFrame({"user": rs.randint(1, 301, n), "item": rs.randint(1, 101, n), "rating": rs.randint(1, 6, n)})
        df = df.drop_duplicates(["user", "item"])
        if name == YahooMusic.TRAIN_FILE:                        # the shape comes from the train file's largest ids (:45-46)
            df = pd.concat([df, pd.DataFrame({"user": [300], "item": [100], "rating": [5]})]).drop_duplicates(["user", "item"], keep="last")
        df.to_csv(d / name, sep="\t", header=False, index=False)
        frames[name] = df
    ds = YahooMusic(root=tmp_path)
    assert ds.train.shape == ds.valid.shape == ds.test.shape == (300, 100)
    tr_pos = frames[YahooMusic.TRAIN_FILE]
    tr_pos = tr_pos[tr_pos.rating >= 4]
    te_pos = frames[YahooMusic.TEST_FILE]
    te_pos = te_pos[te_pos.rating >= 4]
    assert ds.train_size + ds.valid_size == len(tr_pos) and ds.valid_size == int(np.ceil(0.1 * len(tr_pos)))
    assert sorted(zip(te_pos.user - 1, te_pos.item - 1)) == sorted(zip(*ds.test.nonzero()))       # 1-based ids in the files
    tr, va = sk.train_test_split(tr_pos, test_size=0.1, random_state=12345)
    assert sorted(zip(va.user - 1, va.item - 1)) == sorted(zip(*ds.valid.nonzero()))
    assert set(np.unique(ds.train.tocsr().data)) == {1.0}
    ds5 = YahooMusic(min_rating=5.0, root=tmp_path)
    assert ds5.train_size + ds5.valid_size == int((frames[YahooMusic.TRAIN_FILE].rating >= 5).sum())


def test_text8_local_loader(tmp_path):
    import zipfile
    from cymf_amd.dataset import CooccurrrenceDataset, Text8
    with pytest.raises(ValueError):
        Text8(lang="de", root=tmp_path)                          # cymf/dataset/text8.py:28
    with pytest.raises(FileNotFoundError):
        Text8(root=tmp_path)                                     # never downloads
    with pytest.raises(NotImplementedError):
        CooccurrrenceDataset("x", root=tmp_path).vocab_size()    # cymf/dataset/cooccurrence.py:31-32
    rs = np.random.RandomState(2)
    words = [f"w{i}" for i in range(40)]
    text = " ".join(words[min(39, int(z) - 1)] for z in rs.zipf(1.4, 5000))
    with zipfile.ZipFile(tmp_path / "text8.zip", "w") as zf:     # only the archive present: unpacked next to it (:45-46)
        zf.writestr("text8", text)
    ds = Text8(min_count=4, window_size=5, root=tmp_path)
    assert (tmp_path / "text8").exists()
    M, i2w = read_text(str(tmp_path / "text8"), 4, 5)
    assert ds.vocab_size() == len(i2w) == ds.X.shape[0] and (ds.X != M).nnz == 0 and ds.i2w == i2w
    (tmp_path / "ja.text8").write_text(text)
    assert Text8(lang="ja", min_count=4, window_size=5, root=tmp_path).vocab_size() == ds.vocab_size()


def test_word_bounds_balance_pairs():
    rs = np.random.RandomState(0)
    central = rs.zipf(1.3, 200000) % 5000
    for world in (1, 2, 3, 8):
        b = dist.word_bounds(central, 5000, world)
        assert b[0] == 0 and b[-1] == 5000 and len(b) == world + 1 and (np.diff(b) >= 0).all()
        per = [np.sum((central >= b[r]) & (central < b[r + 1])) for r in range(world)]
        assert sum(per) == len(central)
        if world > 1:
            heaviest = np.bincount(central).max()
            assert max(per) <= len(central) / world + heaviest       # off by at most the word on the boundary


def test_jump_polynomial_header_matches_its_generator():
    """csrc/mt_jump_poly.h is generated (tools/gen_mt_jump.py): the committed coefficients are the ones the tool derives --
    phi by Berlekamp-Massey on numpy's MT19937, g_0 = t^J mod phi, g_{l+1} = g_l^4 -- for the J the header states."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_mt_jump", os.path.join(ROOT, "tools", "gen_mt_jump.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = open(os.path.join(ROOT, "cymf_amd", "csrc", "mt_jump_poly.h")).read()
    assert int(re.search(r"MT_JUMP_WORDS = (\d+)LL", text).group(1)) == gen.J == 624 * gen.BLOCKS
    levels = int(re.search(r"MT_JUMP_LEVELS = (\d+)", text).group(1))
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{8})u", text)]
    assert levels == gen.LEVELS and len(words) == levels * 624
    rs = np.random.RandomState(1234)
    x = gen.mt_words(rs.get_state()[1].astype(np.uint32), 2 * gen.DEG + gen.N + 8)
    C, L = gen.berlekamp_massey([x[k] >> 31 for k in range(1, 2 * gen.DEG + 1)])
    assert L == gen.DEG
    phi = sum(((C >> i) & 1) << (L - i) for i in range(L + 1))
    g = gen.power_of_t(gen.J, phi)
    for lvl in range(levels):
        have = sum(w << (32 * i) for i, w in enumerate(words[lvl * 624:(lvl + 1) * 624]))
        assert have == g, lvl
        g = gen.gf2_mod(gen.gf2_square(gen.gf2_mod(gen.gf2_square(g), phi)), phi)


def test_micro_benchmarks_cross_compile(tmp_path):
    """tools/micro/*.hip are the standalone programs behind profiles/r02_lds_atomics.md, r02_permlane_hazard.md and the
    bit-identity claim of wave_sum: they must keep building for gfx950 (hipcc cross-compiles without a GPU)."""
    import glob
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    srcs = sorted(glob.glob(os.path.join(ROOT, "tools", "micro", "*.hip")))
    assert srcs
    for src in srcs:
        out = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def test_epoch_chunks_cover_the_epochs_and_grow_only_while_calls_are_short():
    """cymf_amd._host.EpochChunks (the fit() drivers): the calls add up to num_epochs; every epoch is its own call where each one
    is looked at; otherwise the calls double while they return quickly and stop growing once one takes longer than the target."""
    import time
    from cymf_amd import _host
    assert list(_host.EpochChunks(5, True)) == [1, 1, 1, 1, 1]
    assert list(_host.EpochChunks(0, False)) == []
    fast = list(_host.EpochChunks(100, False, target=10.0, cap=16))          # every call "returns at once"
    assert sum(fast) == 100 and fast[:5] == [1, 2, 4, 8, 16] and max(fast) == 16
    slow = []
    for n in _host.EpochChunks(6, False, target=0.0):                        # no call is ever short enough
        slow.append(n)
        time.sleep(0.001)
    assert slow == [1] * 6
    c = _host.EpochChunks(50, False, target=10.0)
    got = []
    for n in c:
        got.append(n)
        if len(got) == 3:
            c.stop()                                                         # early stopping
    assert got == [1, 2, 4]

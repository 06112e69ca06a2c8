"""Dataset loaders with LOCAL paths only.  The reference's MovieLens loader downloads
with wget (cymf/dataset/movielens.py:31-40); there is no network here, so this one reads the same files
from `root/dir_name` (default ~/.cymf, the reference's cache directory, cymf/dataset/implicit.py:35-37) and
raises FileNotFoundError otherwise.  Everything after the read follows the reference: reset_id
(:76-85), rating >= min_rating -> 1.0 (:59-60), the 90/10 then 90/10 split with random_state=12345
(:62-63), lil_matrix outputs (:65-67)."""
from pathlib import Path

import numpy as np
from scipy import sparse


class ImplicitFeedbackDataset(object):
    """cymf/dataset/implicit.py:19-54."""

    def __init__(self, dir_name, min_rating=4.0, root=None):
        self.root = Path(root) if root is not None else Path.home().joinpath(".cymf")
        self.dir_path = self.root.joinpath(dir_name)
        self.min_rating = min_rating

    def to_matrix(self, df):
        m = sparse.coo_matrix((df["rating"].values.astype(np.float64), (df["user"].values, df["item"].values)),
                              shape=(self.num_user, self.num_item)).tocsr()
        m.data[:] = np.minimum(m.data, 1.0) if len(m.data) else m.data     # duplicates: the reference assigns, not adds
        return m.tolil()


class MovieLens(ImplicitFeedbackDataset):
    """cymf/dataset/movielens.py:24-85, minus the download."""

    def __init__(self, dir_name="ml-100k", min_rating=4.0, root=None):
        super().__init__(dir_name, min_rating, root)
        if dir_name not in ("ml-100k", "ml-1m"):
            raise ValueError("dir_name must be 'ml-100k' or 'ml-1m'.")
        import pandas as pd
        from sklearn.model_selection import train_test_split
        rating_file = self.dir_path.joinpath("u.data" if dir_name == "ml-100k" else "ratings.dat")
        if not rating_file.exists():
            raise FileNotFoundError(f"{rating_file} not found: this loader never downloads; unpack {dir_name}.zip "
                                    f"from grouplens.org under {self.root}")
        if dir_name == "ml-100k":
            df_all = pd.read_csv(rating_file, sep="\t", names=("user", "item", "rating", "timestamp"))
        else:
            df_all = pd.read_csv(rating_file, sep="::", names=("user", "item", "rating", "timestamp"), engine="python")
        df_all["item"] = self.reset_id(df_all["item"])
        df_all["user"] = self.reset_id(df_all["user"])
        self.num_user = len(set(df_all["user"]))
        self.num_item = len(set(df_all["item"]))
        df_all = df_all[df_all["rating"] >= self.min_rating].copy()
        df_all["rating"] = 1.0
        self.df_train, self.df_test = train_test_split(df_all, test_size=0.1, random_state=12345)
        self.df_train, self.df_valid = train_test_split(self.df_train, test_size=0.1, random_state=12345)
        self.train = self.to_matrix(self.df_train)
        self.valid = self.to_matrix(self.df_valid)
        self.test = self.to_matrix(self.df_test)
        self.train_size = self.train.nnz
        self.valid_size = self.valid.nnz
        self.test_size = self.test.nnz

    @staticmethod
    def reset_id(column):
        x2index = {}
        for x in set(column):                      # the reference iterates the set, not the sorted values
            if x not in x2index:
                x2index[x] = len(x2index)
        return column.map(lambda x: x2index[x])


class YahooMusic(ImplicitFeedbackDataset):
    """cymf/dataset/yahoomusic.py:18-61 (Yahoo! R3).  The reference also only reads local files here (:23-26, it prints
    where to get them and exits); this one raises FileNotFoundError instead of ending the interpreter.  Ids are 1-based in
    the files (:31-32, :40-41); the matrix shape comes from the TRAIN file's largest ids (:45-46), so a test pair outside
    it raises in to_matrix exactly as the reference's would; the train file is split 90/10 into train/valid with
    random_state=12345 (:48), the test file is used whole."""

    TRAIN_FILE = "ydata-ymusic-rating-study-v1_0-train.txt"
    TEST_FILE = "ydata-ymusic-rating-study-v1_0-test.txt"

    def __init__(self, min_rating=4.0, under_sampling=None, root=None):
        super().__init__("yahoomusic", min_rating, root)
        import pandas as pd
        from sklearn.model_selection import train_test_split
        frames = []
        for name in (self.TRAIN_FILE, self.TEST_FILE):
            path = self.dir_path.joinpath(name)
            if not path.exists():
                raise FileNotFoundError(f"{path} not found: get the R3 dataset from webscope.sandbox.yahoo.com and put it under {self.dir_path}")
            df = pd.read_csv(path, sep="\t", names=["user", "item", "rating"])
            df["user"] -= 1
            df["item"] -= 1
            df = df[df["rating"] >= min_rating].copy()
            df["rating"] = 1.0
            frames.append(df)
        self.df_train, self.df_test = frames
        self.num_user = int(self.df_train.user.max()) + 1
        self.num_item = int(self.df_train.item.max()) + 1
        self.df_train, self.df_valid = train_test_split(self.df_train, test_size=0.1, random_state=12345)
        self.train = self.to_matrix(self.df_train)
        self.valid = self.to_matrix(self.df_valid)
        self.test = self.to_matrix(self.df_test)
        self.train_size = self.train.nnz
        self.valid_size = self.valid.nnz
        self.test_size = self.test.nnz


class CooccurrrenceDataset(object):
    """cymf/dataset/cooccurrence.py:18-32 (the class name keeps the reference's spelling)."""

    def __init__(self, fname, min_count=5, window_size=10, root=None):
        self.root = Path(root) if root is not None else Path.home().joinpath(".cymf")
        self.path = self.root.joinpath(fname)
        self.min_count = min_count
        self.window_size = window_size

    def vocab_size(self):
        raise NotImplementedError()


class Text8(CooccurrrenceDataset):
    """cymf/dataset/text8.py:20-53, minus the download: reads `root/text8` (lang="en") or `root/ja.text8` (lang="ja"),
    or unpacks `root/<name>.zip` if only the archive is there (:45-46), then builds the co-occurrence matrix with
    cymf_amd.glove.read_text (:48)."""

    def __init__(self, lang="en", min_count=5, window_size=10, root=None):
        if lang not in ("en", "ja"):
            raise ValueError("An argument 'lang' must be 'en' or 'ja'.")
        super().__init__("text8" if lang == "en" else "ja.text8", min_count, window_size, root)
        if not self.path.exists():
            zip_path = self.path.parent.joinpath(self.path.name + ".zip")
            if not zip_path.exists():
                raise FileNotFoundError(f"{self.path} (or {zip_path.name}) not found: this loader never downloads")
            import zipfile
            with zipfile.ZipFile(zip_path) as zf:
                zf.extractall(self.path.parent)
        from .glove import read_text
        self.X, self.i2w = read_text(str(self.path), self.min_count, self.window_size)

    def vocab_size(self):
        return len(self.i2w)

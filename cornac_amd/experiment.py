"""The callers of the hot path: split, fit, evaluate, report — `RatioSplit` / `StratifiedSplit` / `CrossValidation` /
`BaseMethod` / `Experiment` / `Result` / `CVResult` with the reference's interface (cornac/eval_methods/base_method.py:229-845,
ratio_split.py:24-126, stratified_split.py:24-145, cross_validation.py:25-143, cornac/experiment/experiment.py:24-167,
result.py:49-116), so that the reference's example scripts
(examples/first_example.py, examples/bpr_netflix.py, ...) run against this backend by changing the import.

What is kept exactly: the split protocol (`RandomState(seed).permutation(len(data))`: train = head, test = tail,
validation in between), the id maps (global maps filled by train, then test, then validation), the `rng` / test-set
reset before every `evaluate`, metric ordering by name, the `Train (s)` / `Test (s)` columns.  What differs: the
evaluation loops are the batched ones of `cornac_amd.eval`; modalities other than what the models of this path take
(`item_image` for VBPR) are not carried."""
import time
from collections import OrderedDict
from math import ceil

import numpy as np

from . import eval as _eval
from .data import Dataset


class Result:
    """per-model averages (`metric_avg_results`) and per-user values (`metric_user_results`), keyed by metric name"""

    def __init__(self, model_name, metric_avg_results, metric_user_results):
        self.model_name = model_name
        self.metric_avg_results = metric_avg_results
        self.metric_user_results = metric_user_results

    def __str__(self):
        return format_table([self])


def format_table(results):
    """one row per model, one column per metric; 4 decimals like the reference's tables"""
    if not results:
        return ""
    headers = list(results[0].metric_avg_results.keys())
    rows = [[r.model_name] + ["{:.4f}".format(r.metric_avg_results[h]) for h in headers] for r in results]
    cells = [[""] + headers] + rows
    widths = [max(len(row[c]) for row in cells) for c in range(len(headers) + 1)]

    def line(row):
        return " | ".join([row[0].ljust(widths[0])] + [v.rjust(w) for v, w in zip(row[1:], widths[1:])])

    bar = " + ".join("-" * w for w in widths)
    return "\n".join([line(cells[0]), bar] + [line(r) for r in rows]) + "\n"


def _rng(seed):
    # cornac/utils/common.py get_rng: None -> numpy's global generator, int -> a fresh RandomState
    if seed is None:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("{} can not be used to create a numpy.random.RandomState".format(seed))


class BaseMethod:
    def __init__(self, data=None, fmt="UIR", rating_threshold=1.0, seed=None, exclude_unknowns=True, verbose=False,
                 **kwargs):
        self.data, self.fmt = data, fmt
        self.rating_threshold, self.seed, self.exclude_unknowns, self.verbose = rating_threshold, seed, exclude_unknowns, verbose
        self.rng = _rng(seed)
        self.train_set = self.test_set = self.val_set = None
        self.global_uid_map, self.global_iid_map = OrderedDict(), OrderedDict()
        self.item_image = kwargs.get("item_image", None)

    @property
    def total_users(self):
        return len(self.global_uid_map)

    @property
    def total_items(self):
        return len(self.global_iid_map)

    @staticmethod
    def organize_metrics(metrics):
        """(rating metrics, ranking metrics), each sorted by name; a ranking metric given a list of k is expanded"""
        if isinstance(metrics, dict):
            rating, ranking = list(metrics.get("rating", [])), list(metrics.get("ranking", []))
        elif isinstance(metrics, list):
            rating, ranking = [], []
            for mt in metrics:
                if getattr(mt, "type", None) == "rating":
                    rating.append(mt)
                elif hasattr(getattr(mt, "k", None), "__len__"):
                    ranking.extend(mt.__class__(k=k) for k in sorted(set(mt.k)))
                else:
                    ranking.append(mt)
        else:
            raise ValueError("Type of metrics has to be either dict or list!")
        return sorted(rating, key=lambda m: m.name), sorted(ranking, key=lambda m: m.name)

    def build(self, train_data, test_data, val_data=None):
        if train_data is None or len(train_data) == 0:
            raise ValueError("train_data is required but None or empty!")
        if test_data is None or len(test_data) == 0:
            raise ValueError("test_data is required but None or empty!")
        self.global_uid_map.clear()
        self.global_iid_map.clear()
        maps = dict(global_uid_map=self.global_uid_map, global_iid_map=self.global_iid_map, seed=self.seed, fmt=self.fmt)
        self.train_set = Dataset.build(train_data, exclude_unknowns=False, **maps)
        self.test_set = Dataset.build(test_data, exclude_unknowns=self.exclude_unknowns, **maps)
        self.val_set = None
        if val_data is not None and len(val_data) > 0:
            self.val_set = Dataset.build(val_data, exclude_unknowns=self.exclude_unknowns, **maps)
        if self.item_image is not None:
            if hasattr(self.item_image, "build"):   # rows into global item-index order (base_method.py:555-606)
                self.item_image.build(id_map=self.global_iid_map)
            for ds in (self.train_set, self.test_set, self.val_set):
                if ds is not None:
                    ds.item_image = self.item_image
        if self.verbose:
            print("---\nTraining data:\nNumber of users = {}\nNumber of items = {}\nNumber of ratings = {}".format(
                self.train_set.num_users, self.train_set.num_items, self.train_set.num_ratings))
            print("---\nTest data:\nNumber of users = {}\nNumber of items = {}\nNumber of ratings = {}".format(
                len(self.test_set.uid_map), len(self.test_set.iid_map), self.test_set.num_ratings))
            print("---\nTotal users = {}\nTotal items = {}".format(self.total_users, self.total_items))
        return self

    @classmethod
    def from_splits(cls, train_data, test_data, val_data=None, fmt="UIR", rating_threshold=1.0, exclude_unknowns=False,
                    seed=None, verbose=False, **kwargs):
        method = cls(fmt=fmt, rating_threshold=rating_threshold, exclude_unknowns=exclude_unknowns, seed=seed,
                     verbose=verbose, **kwargs)
        return method.build(train_data=train_data, test_data=test_data, val_data=val_data)

    def _eval(self, model, test_set, val_set, rating_metrics, ranking_metrics, user_based):
        avg, per_user = OrderedDict(), OrderedDict()
        a, u = _eval.rating_eval(model, rating_metrics, test_set, user_based=user_based, verbose=self.verbose)
        for mt, av, us in zip(rating_metrics, a, u):
            avg[mt.name], per_user[mt.name] = av, us
        a, u = _eval.ranking_eval(model, ranking_metrics, self.train_set, test_set, val_set=val_set,
                                  rating_threshold=self.rating_threshold, exclude_unknowns=self.exclude_unknowns,
                                  verbose=self.verbose)
        for mt, av, us in zip(ranking_metrics, a, u):
            avg[mt.name], per_user[mt.name] = av, us
        return Result(model.name, avg, per_user)

    def evaluate(self, model, metrics, user_based, show_validation=True):
        """fit on the training set, evaluate on the test (and validation) set -> (test Result, validation Result | None)"""
        if self.train_set is None:
            raise ValueError("train_set is required but None!")
        if self.test_set is None:
            raise ValueError("test_set is required but None!")
        self.rng = _rng(self.seed)
        self.test_set = self.test_set.reset()
        if self.verbose:
            print("\n[{}] Training started!".format(model.name))
        t0 = time.time()
        model.fit(self.train_set, self.val_set)
        train_time = time.time() - t0
        if self.verbose:
            print("\n[{}] Evaluation started!".format(model.name))
        rating_metrics, ranking_metrics = self.organize_metrics(metrics)
        t0 = time.time()
        if hasattr(model, "transform"):
            model.transform(self.test_set)
        test_result = self._eval(model, self.test_set, self.val_set, rating_metrics, ranking_metrics, user_based)
        test_result.metric_avg_results["Train (s)"] = train_time
        test_result.metric_avg_results["Test (s)"] = time.time() - t0
        val_result = None
        if show_validation and self.val_set is not None:
            t0 = time.time()
            if hasattr(model, "transform"):
                model.transform(self.val_set)
            val_result = self._eval(model, self.val_set, None, rating_metrics, ranking_metrics, user_based)
            val_result.metric_avg_results["Time (s)"] = time.time() - t0
        return test_result, val_result


class RatioSplit(BaseMethod):
    def __init__(self, data, test_size=0.2, val_size=0.0, rating_threshold=1.0, seed=None, exclude_unknowns=True,
                 verbose=False, **kwargs):
        super().__init__(data=data, rating_threshold=rating_threshold, seed=seed, exclude_unknowns=exclude_unknowns,
                         verbose=verbose, **kwargs)
        self.train_size, self.val_size, self.test_size = self.validate_size(val_size, test_size,
                                                                            kwargs.get("data_size", len(data)))
        self._split()

    @staticmethod
    def validate_size(val_size, test_size, data_size):
        """fractions (< 1) or absolute counts -> (train, validation, test) counts (ratio_split.py:76-111)"""
        sizes = {}
        for name, size in (("val_size", val_size), ("test_size", test_size)):
            size = 0.0 if size is None else size
            if size < 0:
                raise ValueError("{}={} should be greater than zero".format(name, size))
            if size >= data_size:
                raise ValueError("{}={} should be smaller than data_size={}".format(name, size, data_size))
            sizes[name] = ceil(size * data_size) if size < 1 else size
        held_out = sizes["val_size"] + sizes["test_size"]
        if held_out >= data_size:
            raise ValueError("val_size + test_size ({}) should be smaller than data_size={}".format(held_out, data_size))
        return int(data_size - held_out), int(sizes["val_size"]), int(sizes["test_size"])

    def _split(self):
        order = self.rng.permutation(len(self.data))
        pick = lambda idx: [self.data[i] for i in idx]   # noqa: E731
        train_idx, test_idx = order[: self.train_size], order[-self.test_size:]
        val_idx = order[self.train_size: -self.test_size]
        self.build(train_data=pick(train_idx), test_data=pick(test_idx),
                   val_data=pick(val_idx) if len(val_idx) > 0 else None)


class StratifiedSplit(BaseMethod):
    """per-user (or per-item) ratio split, optionally chronological: the newest ratings of every group are held out
    (cornac/eval_methods/stratified_split.py:24-145; same per-group `validate_size` and `rng.permutation` protocol)"""

    def __init__(self, data, group_by="user", chrono=False, fmt="UIRT", test_size=0.2, val_size=0.0, rating_threshold=1.0,
                 seed=None, exclude_unknowns=True, verbose=False, **kwargs):
        super().__init__(data=data, fmt=fmt, rating_threshold=rating_threshold, seed=seed,
                         exclude_unknowns=exclude_unknowns, verbose=verbose, **kwargs)
        if group_by not in ("user", "item"):
            raise ValueError("group_by option must be either 'user' or 'item' but {}".format(group_by))
        if chrono and (fmt != "UIRT" or len(self.data[0]) != 4):
            raise ValueError('Input data must be in "UIRT" format for sorting chronologically.')
        self.chrono, self.group_by, self.val_size, self.test_size = chrono, group_by, val_size, test_size
        self._split()

    def _split(self):
        data = sorted(self.data, key=lambda t: t[3]) if self.chrono else self.data
        groups = OrderedDict()
        col = 0 if self.group_by == "user" else 1
        for idx, rec in enumerate(data):
            groups.setdefault(rec[col], []).append(idx)
        parts = {"train": [], "val": [], "test": []}
        for members in groups.values():
            n_train, _, n_test = RatioSplit.validate_size(self.val_size, self.test_size, len(members))
            if self.chrono:   # the oldest n_train stay in place, only the held-out tail is shuffled
                members = members[:n_train] + self.rng.permutation(members[n_train:]).tolist()
            else:
                members = self.rng.permutation(members).tolist()
            parts["train"] += members[:n_train]
            parts["test"] += members[-n_test:]
            parts["val"] += members[n_train:-n_test]
        pick = lambda idx: [data[i] for i in idx]   # noqa: E731
        self.build(train_data=pick(parts["train"]), test_data=pick(parts["test"]),
                   val_data=pick(parts["val"]) if parts["val"] else None)


class CVResult(list):
    """the per-fold Results of one model + their mean / standard deviation per metric (result.py:79-116)"""

    def __init__(self, model_name):
        super().__init__()
        self.model_name = model_name
        self.metric_mean, self.metric_std = OrderedDict(), OrderedDict()
        self.table = ""

    def organize(self):
        headers = list(self[0].metric_avg_results.keys())
        values = np.asarray([[r.metric_avg_results[h] for h in headers] for r in self], dtype=float)
        for h, mean, std in zip(headers, values.mean(axis=0), values.std(axis=0)):
            self.metric_mean[h], self.metric_std[h] = mean, std
        rows = [Result("Fold %d" % f, r.metric_avg_results, None) for f, r in enumerate(self)]
        rows += [Result("Mean", self.metric_mean, None), Result("Std", self.metric_std, None)]
        self.table = format_table(rows)

    def __str__(self):
        return "[{}]\n{}".format(self.model_name, self.table)


class CrossValidation(BaseMethod):
    """n-fold cross validation (cornac/eval_methods/cross_validation.py:25-143): the fold label of every rating is
    drawn once (`rng.shuffle` of equal-sized labels + `rng.choice` for the remainder) or given as `partition`; every
    fold trains a `clone()` of the model on the other folds"""

    def __init__(self, data, n_folds=5, rating_threshold=1.0, partition=None, seed=None, exclude_unknowns=True,
                 verbose=False, **kwargs):
        super().__init__(data=data, rating_threshold=rating_threshold, seed=seed, exclude_unknowns=exclude_unknowns,
                         verbose=verbose, **kwargs)
        self.n_folds, self.n_ratings, self.current_fold = n_folds, len(self.data), 0
        if partition is None:
            per_fold = self.n_ratings // n_folds
            partition = np.repeat(np.arange(n_folds), per_fold)
            self.rng.shuffle(partition)
            rest = self.n_ratings - per_fold * n_folds
            if rest > 0:
                partition = np.concatenate((partition, self.rng.choice(n_folds, size=rest, replace=True, p=None)))
        elif len(partition) != self.n_ratings:
            raise ValueError("The partition length must be equal to the number of ratings")
        elif len(set(partition)) != n_folds:
            raise ValueError("Number of folds in given partition different from %s" % n_folds)
        self._partition = np.asarray(partition)

    def _get_train_test(self):
        if self.verbose:
            print("Fold: {}".format(self.current_fold + 1))
        held_out = self._partition == self.current_fold
        train = [self.data[i] for i in np.flatnonzero(~held_out)]
        test = [self.data[i] for i in np.flatnonzero(held_out)]
        self.build(train_data=train, test_data=test, val_data=test)

    def evaluate(self, model, metrics, user_based, show_validation=False):
        result = CVResult(model.name)
        for _ in range(self.n_folds):
            self._get_train_test()
            fold_result, _ = BaseMethod.evaluate(self, model.clone(), metrics, user_based, show_validation=False)
            result.append(fold_result)
            self.current_fold = (self.current_fold + 1) % self.n_folds
        result.organize()
        return result, None


class Experiment:
    """fit and evaluate every model with one evaluation method; `.result` / `.val_result` hold the Result rows"""

    def __init__(self, eval_method, models, metrics, user_based=True, show_validation=True, verbose=False, save_dir=None):
        self.eval_method, self.models, self.metrics = eval_method, list(models), list(metrics)
        self.user_based, self.show_validation, self.verbose, self.save_dir = user_based, show_validation, verbose, save_dir
        self.result, self.val_result = None, None

    def run(self):
        self.result, self.val_result = [], []
        if self.save_dir is not None:
            import os

            os.makedirs(self.save_dir, exist_ok=True)
        for model in self.models:
            test_result, val_result = self.eval_method.evaluate(model=model, metrics=self.metrics,
                                                                user_based=self.user_based,
                                                                show_validation=self.show_validation)
            self.result.append(test_result)
            if val_result is not None:
                self.val_result.append(val_result)
            if self.save_dir is not None and hasattr(model, "save"):
                model.save(self.save_dir)
        output = ""
        if self.val_result:
            output += "\nVALIDATION:\n...\n" + format_table(self.val_result)
        if self.result and isinstance(self.result[0], CVResult):   # one table of folds per model, then the means
            output += "\nTEST:\n...\n" + "\n".join(str(r) for r in self.result)
            output += "\n" + format_table([Result(r.model_name, r.metric_mean, None) for r in self.result])
        else:
            output += "\nTEST:\n...\n" + format_table(self.result)
        print(output)
        if self.save_dir is not None:   # the report next to the saved models (experiment.py:160-167; the reference also
            from datetime import datetime   # writes it to the working directory when no save_dir is given — not done here)

            name = "CornacExp-{}.log".format(datetime.now().strftime("%Y-%m-%d_%H-%M-%S-%f"))
            with open(os.path.join(self.save_dir, name), "w") as f:
                f.write(output)
        return self

"""The reference's training driver (diffpiso/combined_training_integrated.py:7-388) on torch: data frames -> coarse grid ->
unrolled PISO steps with the CNN closure -> losses -> Adam, with the reference's recovery logic for failed linear solves.
Same dictionaries (`physical_parameters`, `simulation_parameters`, `training_dict`) and the same files in `base_dir`
(`model_last_working`, `model_epoch_%06d[i%06d].ckpt`, `training_loss_progression.npz`, `validation_loss_progression.npz`,
`loss.log`); checkpoints are `torch.save`d state lists instead of TF checkpoints, plots are not produced."""
import os

import numpy as np
import torch

from .datamanagement import data_path_assembler, load_function, make_dataset
from .grids import CenteredGrid, StaggeredGrid
from .setups import update_dirichlet_values
from .unroll import run_piso_steps


def boundary_perturbation_fun(domain, average_velocity, shape, time, perturbation_amplitudes):
    """combined_training_integrated.py:7-14 (inflow forcing after Ko et al.): two modes cos(n y) sech^2(y/2) sin(omega t)."""
    size_y = float(domain.box.size[0])
    y = np.linspace(0, size_y, int(domain.resolution[0]) + 2) - size_y / 2
    eps = [perturbation_amplitudes[0] * average_velocity, perturbation_amplitudes[1] * average_velocity]
    n, omega = [.4 * np.pi, .3 * np.pi], [.22, .11]
    u = np.sum([eps[i] * np.cos(n[i] * y) * (1 - np.tanh(y / 2) ** 2) * np.sin(omega[i] * time) for i in range(2)], axis=0)
    return np.reshape(u, shape)


def _save(weights, path):
    torch.save([w.detach().cpu() for w in weights], path)


def _restore(weights, path):
    with torch.no_grad():
        for w, v in zip(weights, torch.load(path)):
            w.copy_(v.to(w.device))


class _Run(object):
    """One differentiable evaluation of `step_count` steps + the loss list of training_dict (:54-70)."""

    def __init__(self, physical_parameters, simulation_parameters, training_dict, solver_precision, buffer_width, sponge_start):
        self.pp, self.sp, self.td = physical_parameters, simulation_parameters, training_dict
        self.sponge_start = sponge_start
        self.buffer_width = buffer_width
        (self.domain, self.sim_physics, self.pressure_solver, self.velocity, self.pressure, self.viscosity_field,
         self.bcx) = simulation_parameters["setup_fun"](simulation_parameters, solver_precision, physical_parameters,
                                                       training_dict["step_count"])
        self.network, self.weights, self.loss_buffer_width = training_dict["network_initialiser"](
            buffer_width=buffer_width, padding=training_dict["padding"])
        self.device = self.velocity.staggered_tensor().device
        self.network.to(self.device)
        self.base_dirichlet = self.sim_physics.dirichlet_values

    def coarse(self, velocity_frame, pressure_frame):
        """HR frame -> simulation grid (:169-170)."""
        v = StaggeredGrid(torch.as_tensor(velocity_frame, device=self.device), self.velocity.box).at(self.velocity)
        p = CenteredGrid(torch.as_tensor(pressure_frame, device=self.device), self.pressure.box).at(self.pressure)
        return (StaggeredGrid(v.staggered_tensor(), self.velocity.box, extrapolation=self.velocity.extrapolation),
                CenteredGrid(p.data, self.pressure.box, self.pressure.extrapolation))

    def forward(self, velocity, pressure, inlet_perturbation=None, step_count=None):
        """The graph-construction call of training_run (:54-56) evaluated eagerly: run_piso_steps with the reference's 14
        arguments; `inlet_perturbation` (one [1,Ny+2,1,1] array per step) stands where the reference feeds bc_placeholders."""
        td = self.td
        wrapper = td.get("network_wrapper")
        if wrapper is None:
            def wrapper(net, nn_in, *_):
                return net(nn_in)
        else:
            # the reference calls neural_network_wrapper(network, input, fluid, physical_parameters, simulation_parameters,
            # loss_buffer_width, buffer_width) (:403, :449); two-argument wrappers (network, input) are accepted as well
            import inspect
            user = wrapper
            if len(inspect.signature(user).parameters) <= 2:
                def wrapper(net, nn_in, *_, _u=user):
                    return _u(net, nn_in)
        placeholders, update = None, None
        if inlet_perturbation is not None:                   # time-dependent inflow (:440-441): bcx + perturbation of step i
            placeholders = torch.as_tensor(np.stack(inlet_perturbation), dtype=torch.float32, device=self.device)
            update = self.sp.get("placeholder_update") or (
                lambda dv, pl: update_dirichlet_values(dv, ((False, False), (True, False)), pl))
        self.sim_physics.dirichlet_values = torch.as_tensor(self.base_dirichlet, dtype=torch.float32, device=self.device)
        td_run = dict(td)
        if step_count is not None:                           # (the roll-out of the model comparison advances one step at a time)
            td_run["step_count"] = step_count
        td_run.setdefault("pressure_included", True)
        td_run.setdefault("loss_influence_range", td["step_count"] + 1)
        out = run_piso_steps(velocity, pressure, self.domain, self.pp, self.sp, td_run, self.network, wrapper,
                             self.sim_physics, self.viscosity_field, self.bcx, placeholders, update, self.loss_buffer_width)
        velocity_all_steps, pressure_all_steps, _, velnew, pnew, _, warn, _, _ = out
        return velocity_all_steps, pressure_all_steps, velnew, pnew, warn

    def loss(self, steps, target):
        td = self.td
        loss = torch.zeros((), device=self.device) if td["sum_steps"] else [torch.zeros((), device=self.device)] * td["step_count"]
        contributions = []
        for fn, factor in zip(td["loss_functions"], td["loss_factor"]):
            loss, contrib = fn(loss, [steps], [target], td["step_count"], self.loss_buffer_width, factor, self.sponge_start,
                               sum_steps=td["sum_steps"], loss_influence_range=td.get("loss_influence_range"))
            contributions.append(float(sum(torch.as_tensor(c).sum() for c in (contrib if isinstance(contrib, list) else [contrib]))))
        total = loss if td["sum_steps"] else sum(loss)
        return total, contributions


def _model_rollout(run, physical_parameters, sp, td, perturb_inlet):
    """The model comparison of combined_training_integrated.py:266-296: from frame start_frame[0] of the first data set,
    `interm_forward_steps` single steps of the solver with the current closure (the reference runs its step_count graph and
    keeps the state after the FIRST step, velocity_all_arrays[0]), then the squared L2 distance to the data frame
    `interm_forward_steps * dx_ratio + start_frame` (dx_ratio, as coded at :274).  Returns that distance (inf if the solver
    produced non-finite values)."""
    from .datamanagement import load_frame
    starting_frame = td["start_frame"][0]
    timesteps = td["interm_forward_steps"]
    vel, prs = run.coarse(load_frame(td["dataset"][0], "velocity", starting_frame), load_frame(td["dataset"][0], "pressure", starting_frame))
    target = run.coarse(load_frame(td["dataset"][0], "velocity", timesteps * sp["dx_ratio"] + starting_frame),
                        load_frame(td["dataset"][0], "pressure", starting_frame))[0].staggered_tensor()
    with torch.no_grad():
        for c in range(timesteps):
            pert = None
            if perturb_inlet:                                                  # :281-289
                time_c = starting_frame * sp["dt"] + sp["dt"] * sp["dt_ratio"] * c
                if "perturbation_temporal_offset" in td:
                    time_c += td["perturbation_temporal_offset"][0]
                pert = [boundary_perturbation_fun(run.domain, physical_parameters["average_velocity"], run.bcx.shape,
                                                  time_c + sp["dt"] * sp["dt_ratio"] * t, td["dataset_characteristics"][0])
                        for t in range(1)]
            _, _, vel, prs, _ = run.forward(vel, prs, pert, step_count=1)
        l2 = float(((target - vel.staggered_tensor()) ** 2).sum())
    return l2 if np.isfinite(l2) else float("inf")


def training_run(base_dir, physical_parameters, simulation_parameters, training_dict, solver_precision=1e-10):
    """combined_training_integrated.py:27-388.  Returns (loss_history, loss_history_validation)."""
    sp, td = simulation_parameters, training_dict
    buffer_width = [[i // sp["dx_ratio"] for i in j] for j in td["HR_buffer_width"]]
    sponge_start = int(sp["HRres"][1] * sp["sponge_ratio"]) // sp["dx_ratio"] if "sponge_ratio" in sp else 0
    perturb_inlet = td.get("perturb_inlet", False)
    learning_rate = td["learning_rate"]
    run = _Run(physical_parameters, sp, td, solver_precision, buffer_width, sponge_start)
    optimizer = torch.optim.Adam(run.weights, lr=learning_rate)

    # ---- data (:96-131)
    start_frames, frame_count, frame_count_test = td["start_frame"], td["frame_count_training"], td["frame_count_validation"]
    if td.get("dataset_characteristics") is not None:
        characteristics = []
        for f in range(len(frame_count)):
            offset = td["perturbation_temporal_offset"][f] if "perturbation_temporal_offset" in td else 0
            characteristics.append([(i * sp["dt"] + offset,) + tuple(td["dataset_characteristics"][f])
                                    for i in range(start_frames[f], start_frames[f] + frame_count[f] + frame_count_test[f])])
    else:
        characteristics = [[(float(i),) for i in range(start_frames[f], start_frames[f] + frame_count[f] + frame_count_test[f])]
                           for f in range(len(frame_count))]
    names = ["velocity", "pressure"]
    steps_per_set = [td["step_count"] for _ in start_frames]
    train_tuple = data_path_assembler(td["dataset"], names, characteristics, start_frame=start_frames, frame_count=frame_count,
                                      step_count=steps_per_set, dt_ratio=sp["dt_ratio"])
    test_chars = [c[frame_count[f]:] for f, c in enumerate(characteristics)]
    test_tuple = data_path_assembler(td["dataset"], names, test_chars,
                                     start_frame=[start_frames[f] + frame_count[f] for f in range(len(frame_count))],
                                     frame_count=frame_count_test, step_count=steps_per_set, dt_ratio=sp["dt_ratio"])
    n_train = sum(frame_count) - len(frame_count) * td["step_count"] * sp["dt_ratio"]
    n_test = sum(frame_count_test) - len(frame_count_test) * td["step_count"] * sp["dt_ratio"]
    loss_history = np.zeros(td["epochs"] * max(n_train, 1))
    loss_history_test = np.zeros(td["epochs"] * max(n_test, 1))
    log = open(os.path.join(base_dir, "loss.log"), "w")
    if td.get("load_model_path") is not None:
        _restore(run.weights, td["load_model_path"])
    restarted, last_epoch_ckpt = False, None
    model_l2_losses, model_descriptors, model_restores = [], [], []          # :141-142 (+ which comparisons rolled the model back)

    def evaluate(sample, train):
        velocity_data, pressure_data, characs = sample
        characs = characs[0] if np.ndim(characs) > 1 else characs
        data_time = float(np.ravel(characs)[0])
        vel, prs = run.coarse(velocity_data[:, 0], pressure_data[:, 0])
        target = torch.stack([run.coarse(velocity_data[:, s], pressure_data[:, s])[0].staggered_tensor()
                              for s in range(1, td["step_count"] + 1)], dim=1)
        pert = None
        if perturb_inlet:
            pert = [boundary_perturbation_fun(run.domain, physical_parameters["average_velocity"], run.bcx.shape,
                                              data_time + sp["dt_ratio"] * t * sp["dt"], np.ravel(characs)[1:])
                    for t in range(td["step_count"])]
        steps, _, _, _, warn = run.forward(vel, prs, pert)
        total, contribs = run.loss(steps, target)
        warned = any(bool(torch.as_tensor(w).any()) for w in warn if w is not None)
        return total, contribs, warned

    for e in range(td["epochs"]):
        first = td.get("start_first_epoch_at", 0) if e == 0 else 0
        for i, sample in enumerate(make_dataset(train_tuple, load_function, batch_size=1, shuffle=True, seed=td.get("seed", e))):
            if i < first or i >= n_train:
                continue
            for group in optimizer.param_groups:
                group["lr"] = learning_rate
            optimizer.zero_grad()
            total, contribs, warned = evaluate(sample, True)
            loss_out = float(total)
            if not warned:                                                       # :190-198
                restarted = False
                total.backward()
                if i % 100 == 0:
                    _save(run.weights, os.path.join(base_dir, "model_last_working"))
                    np.savez(os.path.join(base_dir, "training_loss_progression"), loss_history)
                if all(w.grad is not None and torch.isfinite(w.grad).all() for w in run.weights):
                    optimizer.step()
            else:                                                                # :199-257
                if restarted and (model_descriptors or last_epoch_ckpt is not None):
                    # second warning in a row (:200-250): back to the last intermediate checkpoint (model_descriptors[-1]); the
                    # reference crashes here before the first one exists - the last epoch checkpoint stands in
                    _restore(run.weights, os.path.join(base_dir, "model_epoch_%s.ckpt" % model_descriptors[-1])
                             if model_descriptors else last_epoch_ckpt)
                elif os.path.exists(os.path.join(base_dir, "model_last_working")):
                    print("RESTARTING FROM LAST WORKING")
                    _restore(run.weights, os.path.join(base_dir, "model_last_working"))
                optimizer = torch.optim.Adam(run.weights, lr=learning_rate)      # adam_reinit
                restarted = True
                loss_out = -1
            msg = "epoch %d  iteration %d  loss: %s warn:%s  loss_contribs %s" % (e, i, loss_out, warned, contribs)
            print(msg)
            log.write(msg + "\n")
            loss_history[e * max(n_train, 1) + i] = loss_out
            interm = td.get("store_interm_ckpts", 0)
            if interm and i > 0 and i % max((n_train - first) // interm, 1) == 0:  # :263-264
                descriptor = "%06di%06d" % (e, i)
                last_epoch_ckpt = os.path.join(base_dir, "model_epoch_%s.ckpt" % descriptor)
                _save(run.weights, last_epoch_ckpt)
                if td.get("interm_forward_steps"):                               # :266-303
                    l2 = _model_rollout(run, physical_parameters, sp, td, perturb_inlet)
                    model_l2_losses.append(l2)
                    model_descriptors.append(descriptor)
                    msg = "model comparison after %d timesteps: %s  l2 %s" % (td["interm_forward_steps"], descriptor, l2)
                    print(msg)
                    log.write(msg + "\n")
                    # "20 x worse than the previous checkpoint => restore the previous one" (:301-303, from the third comparison on)
                    if len(model_l2_losses) > 2 and model_l2_losses[-1] > 20 * model_l2_losses[-2]:
                        _restore(run.weights, os.path.join(base_dir, "model_epoch_%s.ckpt" % model_descriptors[-2]))
                        model_restores.append((descriptor, model_descriptors[-2]))
                        print("MODEL COMPARISON: restored model_epoch_%s.ckpt" % model_descriptors[-2])
                    np.savez(os.path.join(base_dir, "model_comparison"), descriptors=np.array(model_descriptors),
                             l2=np.array(model_l2_losses), restores=np.array(model_restores, dtype=str).reshape(-1, 2))
        with torch.no_grad():                                                    # validation (:306-330)
            for i, sample in enumerate(make_dataset(test_tuple, load_function, batch_size=1, shuffle=False)):
                if i >= n_test:
                    break
                total, _, _ = evaluate(sample, False)
                print("epoch %d  validation %d  validation_loss: %s" % (e, i, float(total)))
                loss_history_test[e * max(n_test, 1) + i] = float(total)
        last_epoch_ckpt = os.path.join(base_dir, "model_epoch_%06d.ckpt" % e)
        _save(run.weights, last_epoch_ckpt)
        if td.get("lr_decay_fun") is not None:
            learning_rate = td["lr_decay_fun"](learning_rate)
    np.savez(os.path.join(base_dir, "training_loss_progression"), loss_history)
    np.savez(os.path.join(base_dir, "validation_loss_progression"), loss_history_test)
    log.close()
    return loss_history, loss_history_test
